"""Two-stage pipeline for beam-search decoding of large batches: the device front end (forward +
log-softmax + top-k, ``hctr_beam_frontend``) of chunk i+1 runs while the host prefix search
(``hctr_beam_search``, all cores) works on chunk i. Both are ctypes calls that release the GIL; the
engine context is used by one thread only (front end), the search needs no context."""
import queue
import threading


def chunk_schedule(n, chunk, tail=16):
    """[(lo, hi)] line ranges: chunks of ``chunk`` lines, the last ones halving down to ``tail``. The device front end is
    more efficient on large chunks (fixed host work per pass), but nothing overlaps the host search of the LAST chunk, so
    the schedule ends on small ones (measured on config 5, 256 lines: search of 64 lines 44 ms, of 16 lines ~18 ms = one
    line's search on one thread)."""
    taper, c = [], chunk
    while c // 2 >= tail:
        c //= 2
        taper.append(c)
    if taper:
        taper.append(taper[-1])                               # the smallest size twice: [.., 32, 16, 16]
    while taper and sum(taper) > n:
        taper.pop(0)
    rem = n - sum(taper)
    front = []
    if rem > 0:                                               # the lines in front of the taper, in balanced chunks
        k = -(-rem // chunk)
        front = [rem // k + 1] * (rem % k) + [rem // k] * (k - rem % k)
    sizes = front + taper
    out, lo = [], 0
    for s in sizes:
        out.append((lo, lo + s))
        lo += s
    return out


def recognize_beam(model, codec, images, widths=None, chunk=64, taper=True, stats=None):
    """Beam-decode ``images`` (uint8 [B,128,W] numpy array or torch tensor, optionally per-line widths)
    with ``codec``'s beam settings. Returns the decoded strings in input order. Every chunk is padded /
    processed exactly like a batch of its own (same results as calling the two stages back to back).
    ``stats`` (optional dict) receives per-chunk wall times: "frontend_ms", "search_ms" and "consumer_wait_ms"."""
    import time
    n = int(images.shape[0])
    if stats is not None:
        stats.update({"chunks": [], "frontend_ms": [], "search_ms": [], "consumer_wait_ms": []})
    spans = chunk_schedule(n, chunk) if taper else [(lo, min(n, lo + chunk)) for lo in range(0, n, chunk)]
    k = min(int(codec.search_depth), int(model.noutput))
    q = queue.Queue(maxsize=2)
    err = []
    stop = threading.Event()

    def producer():
        try:
            for lo, hi in spans:
                if stop.is_set():
                    break
                wd = None if widths is None else widths[lo:hi]
                t0 = time.perf_counter()
                fe = model.beam_frontend(images[lo:hi], k=k, widths=wd, want_candidates=codec.skip_search)
                if stats is not None:
                    stats["chunks"].append(hi - lo)
                    stats["frontend_ms"].append(round((time.perf_counter() - t0) * 1e3, 2))
                while not stop.is_set():
                    try:
                        q.put((lo, fe), timeout=0.1)
                        break
                    except queue.Full:
                        pass
        except BaseException as exc:      # surfaced in the consumer thread
            err.append(exc)
        finally:
            while True:                   # the end marker must get through even if the consumer has stopped reading
                try:
                    q.put(None, timeout=0.1)
                    break
                except queue.Full:
                    if stop.is_set():
                        break

    t = threading.Thread(target=producer, daemon=True)
    t.start()
    out = [None] * n
    try:
        while True:
            t0 = time.perf_counter()
            item = q.get()
            t1 = time.perf_counter()
            if item is None:
                break
            lo, fe = item
            for i, text in enumerate(codec.decode_frontend(fe)):
                out[lo + i] = text
            if stats is not None:
                stats["consumer_wait_ms"].append(round((t1 - t0) * 1e3, 2))
                stats["search_ms"].append(round((time.perf_counter() - t1) * 1e3, 2))
    finally:
        # an exception in the search (e.g. the reference-compatible IndexError of an empty line) must not leave
        # the producer running on the model's single-threaded engine context: stop it, unblock it, wait for it
        stop.set()
        while t.is_alive():
            try:
                q.get_nowait()
            except queue.Empty:
                pass
            t.join(timeout=0.05)
    if err:
        raise err[0]
    return out
