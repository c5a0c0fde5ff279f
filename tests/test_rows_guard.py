"""The readers-writer guard of HipFlatVectorStore (searches shared, add_texts / delete exclusive): no GPU needed."""
import threading
import time

from rag_arc_amd.encapsulation.database.vector_db.hip_flat import _RowsGuard


def _run(fn):
    t = threading.Thread(target=fn, daemon=True)
    t.start()
    return t


def test_readers_share_and_a_writer_excludes_them():
    g, log, inside = _RowsGuard(), [], threading.Event()

    def reader(name):
        with g.shared():
            log.append(("in", name))
            inside.wait(2)
            log.append(("out", name))

    r1, r2 = _run(lambda: reader("a")), _run(lambda: reader("b"))
    time.sleep(0.05)
    assert sorted(x for x in log) == [("in", "a"), ("in", "b")]          # both inside at once

    def writer():
        with g.exclusive():
            log.append(("w", None))

    w = _run(writer)
    time.sleep(0.05)
    assert ("w", None) not in log                                        # ... and the writer waits for them
    inside.set()
    for t in (r1, r2, w):
        t.join(2)
        assert not t.is_alive()
    assert log[-1] == ("w", None)


def test_a_nested_search_of_a_reader_does_not_wait_for_the_writer_queued_behind_it():
    g, got_outer, release, done = _RowsGuard(), threading.Event(), threading.Event(), []

    def reader():
        with g.shared():
            got_outer.set()
            release.wait(2)
            with g.shared():              # (a queued writer has priority over NEW readers, not over this one)
                done.append("nested")
        done.append("reader out")

    def writer():
        got_outer.wait(2)
        with g.exclusive():
            done.append("writer")

    r, w = _run(reader), _run(writer)
    got_outer.wait(2)
    time.sleep(0.05)                      # the writer is queued now
    assert g.waiting_writers == 1
    release.set()
    for t in (r, w):
        t.join(2)
        assert not t.is_alive(), "deadlock"
    assert done == ["nested", "reader out", "writer"]


def test_the_writer_is_reentrant_and_may_search():
    g = _RowsGuard()
    with g.exclusive():
        with g.exclusive():               # delete -> rebuild -> add_texts
            with g.shared():              # ... -> a search of its own
                assert g.depth == 3 and g.readers == 0
        assert g.writer is not None
    assert g.writer is None and g.depth == 0
    with g.shared():                      # and the guard is free again
        assert g.readers == 1
    assert g.readers == 0


def test_new_readers_queue_behind_a_waiting_writer():
    g, first_in, let_go, order = _RowsGuard(), threading.Event(), threading.Event(), []

    def first():
        with g.shared():
            first_in.set()
            let_go.wait(2)

    def writer():
        with g.exclusive():
            order.append("writer")

    def late_reader():
        with g.shared():
            order.append("late reader")

    a = _run(first)
    first_in.wait(2)
    w = _run(writer)
    time.sleep(0.05)
    b = _run(late_reader)
    time.sleep(0.05)
    assert order == []
    let_go.set()
    for t in (a, w, b):
        t.join(2)
        assert not t.is_alive()
    assert order == ["writer", "late reader"]
