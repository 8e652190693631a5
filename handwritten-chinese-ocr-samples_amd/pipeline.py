"""Two-stage pipeline for beam-search decoding of large batches: the device front end (forward +
log-softmax + top-k, ``hctr_beam_frontend``) of chunk i+1 runs while the host prefix search
(``hctr_beam_search``, all cores) works on chunk i. Both are ctypes calls that release the GIL; the
engine context is used by one thread only (front end), the search needs no context."""
import queue
import threading


def recognize_beam(model, codec, images, widths=None, chunk=32):
    """Beam-decode ``images`` (uint8 [B,128,W] numpy array or torch tensor, optionally per-line widths)
    with ``codec``'s beam settings. Returns the decoded strings in input order. Every chunk is padded /
    processed exactly like a batch of its own (same results as calling the two stages back to back)."""
    n = int(images.shape[0])
    k = min(int(codec.search_depth), int(model.noutput))
    q = queue.Queue(maxsize=2)
    err = []
    stop = threading.Event()

    def producer():
        try:
            for lo in range(0, n, chunk):
                if stop.is_set():
                    break
                hi = min(n, lo + chunk)
                wd = None if widths is None else widths[lo:hi]
                fe = model.beam_frontend(images[lo:hi], k=k, widths=wd, want_candidates=codec.skip_search)
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
            item = q.get()
            if item is None:
                break
            lo, fe = item
            for i, text in enumerate(codec.decode_frontend(fe)):
                out[lo + i] = text
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
