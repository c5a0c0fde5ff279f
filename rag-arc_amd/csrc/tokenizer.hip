// tokenizer.hip — host-side batch WordPiece tokenisation (C-ABI: rarc_wordpiece_create / _encode / _destroy).  No device code.
//
// Behind `SentenceTransformer.encode(texts)` (core/file_management/embeddings/huggingface.py:122-126) the reference's
// texts pass through the BERT tokeniser of the `tokenizers` library — native (Rust), multi-threaded — before the model
// sees them.  The python restatement in encapsulation/embeddings/wordpiece.py (pinned token for token to
// transformers.BertTokenizer) does ~0.1-0.25 M tokens/s on one core; the MI355X encoder consumes 3-10 M tokens/s.  This
// is the same algorithm for the texts an English corpus is made of — pure ASCII — on n_threads cores:
//
//   special tokens cut out of the raw text wherever they occur  ->  between them: clean-up (drop NUL / control characters,
//   whitespace -> ' ')  ->  split on whitespace  ->  lower-case (uncased vocabularies)  ->  every punctuation character its own token  ->  greedy
//   longest-match-first WordPiece with "##" continuations ([UNK] for a word that cannot be covered or is longer than
//   max_input_chars_per_word)  ->  [CLS] ids[: max_length - 2] [SEP]
//
// For ASCII the Unicode steps of the python version (NFC, NFD + accent stripping, CJK spacing, full-Unicode lower-casing)
// are identities, so the two agree exactly; a text with any byte >= 0x80 is NOT tokenised here: its length comes back
// as -1 and the caller runs the python tokeniser on it (wordpiece.WordPieceTokenizer.encode_batch does).
#include <string.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "rarc_common.h"

struct RarcWordPiece {
  std::vector<std::string> tokens;                          // id -> token (owns the bytes the maps point into)
  std::unordered_map<std::string, int32_t> head, tail;      // whole-word pieces; "##" continuations stored WITHOUT the "##"
  std::vector<std::string> specials;                        // cut out of the raw text wherever they occur; longest first
  std::vector<int32_t> special_id;
  bool special_first[256] = {};                             // first bytes of the specials
  int32_t unk = 0, cls = 0, sep = 0, pad = 0;
  int max_chars = 100;
  bool lower = true;
  size_t longest_head = 0, longest_tail = 0;
};

namespace {

inline bool is_punct(unsigned char c) { return (c >= 33 && c <= 47) || (c >= 58 && c <= 64) || (c >= 91 && c <= 96) || (c >= 123 && c <= 126); }
inline bool is_space(unsigned char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }
inline bool is_control(unsigned char c) { return (c < 32 && c != '\t' && c != '\n' && c != '\r') || c == 127; }   // category Cc

// one text -> ids (without specials), at most `room` of them; false = the text is not ASCII
bool encode_ascii(const RarcWordPiece& wp, const char* text, int64_t n, int room, std::vector<int32_t>& out, std::string& word,
                  std::string& sub) {
  out.clear();
  for (int64_t i = 0; i < n; ++i)
    if ((unsigned char)text[i] >= 0x80) return false;
  auto wordpiece = [&](const char* w, size_t len) {   // greedy longest match first over one punctuation-free word
    if ((int)out.size() >= room) return;
    if (len > (size_t)wp.max_chars) { out.push_back(wp.unk); return; }
    const size_t first = out.size();
    size_t start = 0;
    while (start < len) {
      const auto& map = start == 0 ? wp.head : wp.tail;
      const size_t longest = start == 0 ? wp.longest_head : wp.longest_tail;
      size_t end = len - start > longest ? start + longest : len;
      int32_t id = -1;
      for (; end > start; --end) {
        sub.assign(w + start, end - start);
        const auto it = map.find(sub);
        if (it != map.end()) { id = it->second; break; }
      }
      if (id < 0) {                        // the word cannot be covered: the WHOLE word is [UNK]
        out.resize(first);
        out.push_back(wp.unk);
        return;
      }
      out.push_back(id);
      start = end;
    }
  };
  // special tokens are cut out of the RAW text first (anywhere, case-sensitive, longest match at the earliest position)
  auto special_at = [&](int64_t pos) -> int {
    if (!wp.special_first[(unsigned char)text[pos]]) return -1;
    for (size_t s = 0; s < wp.specials.size(); ++s) {      // sorted longest first
      const std::string& t = wp.specials[s];
      if ((int64_t)t.size() <= n - pos && memcmp(text + pos, t.data(), t.size()) == 0) return (int)s;
    }
    return -1;
  };
  auto segment = [&](int64_t i, const int64_t e) {   // ordinary text [i, e): the BERT tokeniser
    while (i < e && (int)out.size() < room) {
      // next whitespace-delimited token of the cleaned text (control characters vanish WITHOUT splitting, as in _clean)
      word.clear();
      while (i < e && (is_space((unsigned char)text[i]) || is_control((unsigned char)text[i]))) ++i;
      for (; i < e; ++i) {
        const unsigned char c = (unsigned char)text[i];
        if (is_space(c)) break;
        if (is_control(c)) continue;
        word.push_back((char)c);
      }
      if (word.empty()) continue;
      if (wp.lower)
        for (auto& ch : word)
          if (ch >= 'A' && ch <= 'Z') ch = (char)(ch + 32);
      size_t a = 0;
      const size_t L = word.size();
      while (a < L && (int)out.size() < room) {
        size_t b = a + 1;
        if (!is_punct((unsigned char)word[a]))            // a punctuation character is a token of its own
          while (b < L && !is_punct((unsigned char)word[b])) ++b;
        wordpiece(word.data() + a, b - a);
        a = b;
      }
    }
  };
  int64_t seg0 = 0;
  for (int64_t pos = 0; pos < n && (int)out.size() < room;) {
    const int sp = wp.specials.empty() ? -1 : special_at(pos);
    if (sp < 0) { ++pos; continue; }
    segment(seg0, pos);
    if ((int)out.size() < room) out.push_back(wp.special_id[(size_t)sp]);
    pos += (int64_t)wp.specials[(size_t)sp].size();
    seg0 = pos;
  }
  segment(seg0, n);
  if ((int)out.size() > room) out.resize(room);
  return true;
}

}  // namespace

extern "C" int rarc_wordpiece_create(const char* vocab_blob, size_t blob_bytes, int do_lower_case, const char* unk_token,
                                     const char* cls_token, const char* sep_token, const char* pad_token,
                                     const char* never_split_blob, size_t never_split_bytes, int max_input_chars_per_word,
                                     RarcWordPiece** out) {
  RARC_REQUIRE(vocab_blob && unk_token && cls_token && sep_token && pad_token && out, RARC_E_INVALID, "rarc_wordpiece_create: null argument");
  RARC_REQUIRE(max_input_chars_per_word > 0, RARC_E_INVALID, "rarc_wordpiece_create: max_input_chars_per_word must be positive");
  *out = nullptr;
  RarcWordPiece* wp = nullptr;
  try {      // (vocabulary tables: allocation failures must not unwind through extern "C")
  wp = new RarcWordPiece();
  wp->lower = do_lower_case != 0;
  wp->max_chars = max_input_chars_per_word;
  size_t a = 0;
  while (a <= blob_bytes) {                 // tokens separated by '\n', id = position (vocab.txt)
    size_t b = a;
    while (b < blob_bytes && vocab_blob[b] != '\n') ++b;
    if (b == blob_bytes && a == blob_bytes) break;
    wp->tokens.emplace_back(vocab_blob + a, b - a);
    a = b + 1;
  }
  for (size_t id = 0; id < wp->tokens.size(); ++id) {
    const std::string& t = wp->tokens[id];
    wp->head[t] = (int32_t)id;              // (later duplicates win, as in the python dict built from the file)
    if (t.size() > wp->longest_head) wp->longest_head = t.size();
    if (t.size() > 2 && t[0] == '#' && t[1] == '#') {
      wp->tail[t.substr(2)] = (int32_t)id;
      if (t.size() - 2 > wp->longest_tail) wp->longest_tail = t.size() - 2;
    }
  }
  auto need = [&](const char* tok, int32_t* id) {
    const auto it = wp->head.find(tok);
    if (it == wp->head.end()) return false;
    *id = it->second;
    return true;
  };
  if (!need(unk_token, &wp->unk) || !need(cls_token, &wp->cls) || !need(sep_token, &wp->sep)) {
    delete wp;
    rarc_set_error("rarc_wordpiece_create: the vocabulary lacks a special token (%s / %s / %s)", unk_token, cls_token, sep_token);
    return RARC_E_INVALID;
  }
  if (!need(pad_token, &wp->pad)) wp->pad = 0;
  // the special tokens: '\n'-separated in special_blob (the caller lists unk / cls / sep / pad / mask and any further token
  // to be matched raw); those missing from the vocabulary are ignored
  for (size_t p = 0; never_split_blob && p < never_split_bytes;) {
    size_t q = p;
    while (q < never_split_bytes && never_split_blob[q] != '\n') ++q;
    if (q > p) {
      std::string t(never_split_blob + p, q - p);
      const auto it = wp->head.find(t);
      if (it != wp->head.end()) { wp->specials.push_back(t); wp->special_id.push_back(it->second); }
    }
    p = q + 1;
  }
  {  // longest first (ties: byte order), as the python side sorts them
    std::vector<size_t> order(wp->specials.size());
    for (size_t k = 0; k < order.size(); ++k) order[k] = k;
    std::sort(order.begin(), order.end(), [&](size_t x, size_t y) {
      const std::string &a = wp->specials[x], &b = wp->specials[y];
      return a.size() != b.size() ? a.size() > b.size() : a < b;
    });
    std::vector<std::string> sp;
    std::vector<int32_t> id;
    for (size_t k : order) { sp.push_back(wp->specials[k]); id.push_back(wp->special_id[k]); }
    wp->specials.swap(sp);
    wp->special_id.swap(id);
    for (const auto& t : wp->specials) wp->special_first[(unsigned char)t[0]] = true;
  }
  } catch (const std::exception& ex) {
    delete wp;
    rarc_set_error("rarc_wordpiece_create: %s", ex.what());
    return RARC_E_INVALID;
  }
  *out = wp;
  return RARC_OK;
}

extern "C" void rarc_wordpiece_destroy(RarcWordPiece* wp) { delete wp; }

extern "C" int rarc_wordpiece_encode(const RarcWordPiece* wp, const char* text_blob, const int64_t* text_offsets, int n_texts,
                                     int max_length, int32_t* h_ids, int64_t ld_ids, int32_t* h_lens, int n_threads) {
  RARC_REQUIRE(wp && text_offsets && h_ids && h_lens && (text_blob || n_texts == 0), RARC_E_INVALID, "rarc_wordpiece_encode: null argument");
  RARC_REQUIRE(n_texts >= 0 && max_length >= 2 && ld_ids >= max_length, RARC_E_INVALID,
               "rarc_wordpiece_encode: need max_length >= 2 and a row stride >= max_length");
  RARC_REQUIRE(n_threads >= 1 && n_threads <= 256, RARC_E_INVALID, "rarc_wordpiece_encode: n_threads must be 1..256");
  std::atomic<int> next{0};
  std::atomic<int> oom{0};
  constexpr int BLOCK = 32;
  auto work = [&]() {
    try {
    std::vector<int32_t> ids;
    std::string word, sub;
    ids.reserve((size_t)max_length);
    for (;;) {
      const int i0 = next.fetch_add(BLOCK);
      if (i0 >= n_texts) break;
      const int i1 = i0 + BLOCK < n_texts ? i0 + BLOCK : n_texts;
      for (int i = i0; i < i1; ++i) {
        int32_t* row = h_ids + (size_t)i * (size_t)ld_ids;
        const bool ok = encode_ascii(*wp, text_blob + text_offsets[i], text_offsets[i + 1] - text_offsets[i], max_length - 2, ids, word, sub);
        if (!ok) {
          h_lens[i] = -1;                   // not ASCII: the caller tokenises this one in python
          continue;
        }
        row[0] = wp->cls;
        memcpy(row + 1, ids.data(), ids.size() * sizeof(int32_t));
        row[1 + ids.size()] = wp->sep;
        const int len = (int)ids.size() + 2;
        for (int j = len; j < max_length; ++j) row[j] = wp->pad;
        h_lens[i] = len;
      }
    }
    } catch (const std::exception&) {
      oom.store(1);
    }
  };
  const int nt = n_texts < n_threads * BLOCK ? (n_texts + BLOCK - 1) / BLOCK : n_threads;
  if (nt <= 1) {
    work();
  } else {
    std::vector<std::thread> th;
    bool started_all = true;
    try {
      th.reserve((size_t)nt);
      for (int t = 0; t < nt; ++t) th.emplace_back(work);
    } catch (const std::exception&) {
      started_all = false;          // (EAGAIN under a thread limit: the threads that did start finish the work)
    }
    if (!started_all && th.empty()) work();
    for (auto& t : th) t.join();
  }
  if (oom.load()) {
    rarc_set_error("rarc_wordpiece_encode: out of host memory");
    return RARC_E_INVALID;
  }
  return RARC_OK;
}
