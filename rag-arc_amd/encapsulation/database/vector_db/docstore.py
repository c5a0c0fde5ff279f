"""Row-indexed docstores: what turns the row numbers of a search answer back into `Document`s.

The reference keeps two dicts, `index_to_docstore_id[row] -> id` and `docstore[id] -> Document`
(encapsulation/database/vector_db/VectorStore_Faiss.py:96-97, looked up per hit at :270-271).  HipFlatVectorStore keeps those
(same names, same pickle) and, next to them, a row-indexed *sequence* — the form a 256 x 100 answer is mapped through in one
native call (hip/hostmap.py).  Two sequences exist:

* a plain python list of the very Document objects the dicts hold (stores built through add_texts);
* `ColumnarDocstore` below for corpus-scale stores (BASELINE configs 4 / 5: 100M rows): contents and ids in two byte blobs
  with offset arrays — 100M python Documents would be ~40 GB of objects and minutes of construction, the columns are what the
  texts weigh plus 16 bytes a row, pickle as four numpy arrays, and a Document is built when a search names its row.
"""
from typing import Any, Dict, Iterable, List, Optional, Sequence

import numpy as np

from ....core.utils.data_model import Document


class ColumnarDocstore:
    """Immutable row -> Document sequence over byte columns.  `metadatas`: None (every document gets {}), or a sequence of
    dicts indexed by row."""

    def __init__(self, text_blob, text_off, id_blob=None, id_off=None, metadatas: Optional[Sequence[dict]] = None):
        self.text_blob = np.ascontiguousarray(text_blob, dtype=np.uint8)
        self.text_off = np.ascontiguousarray(text_off, dtype=np.int64)
        if self.text_off.ndim != 1 or self.text_off.size < 1 or int(self.text_off[-1]) > self.text_blob.size:
            raise ValueError("text offsets must be [n + 1] and end inside the blob")
        self.id_blob = None if id_blob is None else np.ascontiguousarray(id_blob, dtype=np.uint8)
        self.id_off = None if id_off is None else np.ascontiguousarray(id_off, dtype=np.int64)
        if (self.id_blob is None) != (self.id_off is None):
            raise ValueError("id blob and id offsets go together")
        if self.id_off is not None and self.id_off.size != self.text_off.size:
            raise ValueError("one id per text")
        if metadatas is not None and len(metadatas) != len(self):
            raise ValueError("one metadata dict per text")
        self.metadatas = metadatas
        self._tmem, self._imem = memoryview(self.text_blob), None if self.id_blob is None else memoryview(self.id_blob)
        self._row_of_id: Optional[Dict[str, int]] = None

    # -- construction -----------------------------------------------------------------------------------------------
    @classmethod
    def from_texts(cls, texts: Iterable[str], ids: Optional[Iterable[str]] = None,
                   metadatas: Optional[Sequence[dict]] = None) -> "ColumnarDocstore":
        def column(strings):
            enc = [s.encode("utf-8") for s in strings]
            off = np.zeros(len(enc) + 1, dtype=np.int64)
            np.cumsum([len(b) for b in enc], out=off[1:])
            return np.frombuffer(b"".join(enc), dtype=np.uint8), off

        tb, to = column(texts)
        ib, io = column(ids) if ids is not None else (None, None)
        return cls(tb, to, ib, io, metadatas)

    @classmethod
    def decimal(cls, n: int, width: int = 0) -> "ColumnarDocstore":
        """Rows 0..n-1 whose content and id are the row number in decimal, zero-padded to `width` digits (default: as many
        as n - 1 needs) — the synthetic corpus of SURVEY.md §8(d) ("content key := decimal id string"), built with numpy."""
        n = int(n)
        width = max(int(width), len(str(max(n - 1, 0))))
        digits = np.empty((n, width), dtype=np.uint8)
        v = np.arange(n, dtype=np.int64)
        for col in range(width - 1, -1, -1):
            digits[:, col] = (v % 10) + 48
            v //= 10
        off = np.arange(n + 1, dtype=np.int64) * width
        return cls(digits.reshape(-1), off)            # (no id column: the id of a row is its content)

    # -- the sequence protocol ----------------------------------------------------------------------------------------
    def __len__(self) -> int:
        return int(self.text_off.size - 1)

    def __getitem__(self, row: int) -> Document:
        row = int(row)
        if row < 0:
            row += len(self)
        if not (0 <= row < len(self)):
            raise IndexError(row)
        a, b = self.text_off[row], self.text_off[row + 1]
        content = str(self._tmem[a:b], "utf-8")
        if self._imem is None:
            doc_id = content
        else:
            a, b = self.id_off[row], self.id_off[row + 1]
            doc_id = str(self._imem[a:b], "utf-8")
        meta = {} if self.metadatas is None else self.metadatas[row]
        return Document(content=content, metadata=meta, id=doc_id)

    def columns(self):
        """What csrc/hostmap.c builds Documents from without calling back into python (see its `Columns`)."""
        return (Document, self.text_blob, self.text_off, self.id_blob, self.id_off, self.metadatas)

    def row_of(self, doc_id: str) -> Optional[int]:
        """Row of the document with this id (None if there is none); builds an id -> row dict on first use."""
        if self._row_of_id is None:
            self._row_of_id = {self[r].id: r for r in range(len(self))}
        return self._row_of_id.get(doc_id)

    # -- pickle: the columns, not the views ---------------------------------------------------------------------------
    def __getstate__(self) -> Dict[str, Any]:
        return {"text_blob": self.text_blob, "text_off": self.text_off, "id_blob": self.id_blob, "id_off": self.id_off,
                "metadatas": self.metadatas}

    def __setstate__(self, state: Dict[str, Any]) -> None:
        self.__init__(state["text_blob"], state["text_off"], state["id_blob"], state["id_off"], state["metadatas"])


def rows_from_dicts(docstore: dict, index_to_docstore_id: dict) -> List[Document]:
    """The row-indexed list equivalent to the reference's two dicts (rows 0..n-1 must all be present)."""
    n = len(index_to_docstore_id)
    try:
        return [docstore[index_to_docstore_id[r]] for r in range(n)]
    except KeyError as exc:
        raise KeyError(f"index_to_docstore_id / docstore do not cover rows 0..{n - 1}: {exc}") from None
