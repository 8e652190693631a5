"""Two-stage pipeline for beam-search decoding of large batches: the device front end (forward +
log-softmax + top-k, ``hctr_beam_frontend``) of chunk i+1 runs while the host prefix search
(``hctr_beam_search``, all cores) works on chunk i. Both are ctypes calls that release the GIL; the
engine context is used by one thread only (front end), the search needs no context."""
import queue
import threading


def recognize_beam(model, codec, images, widths=None, chunk=64):
    """Beam-decode ``images`` (uint8 [B,128,W] numpy array or torch tensor, optionally per-line widths)
    with ``codec``'s beam settings. Returns the decoded strings in input order. Every chunk is padded /
    processed exactly like a batch of its own (same results as calling the two stages back to back)."""
    n = int(images.shape[0])
    k = min(int(codec.search_depth), int(model.noutput))
    q = queue.Queue(maxsize=2)
    err = []

    def producer():
        try:
            for lo in range(0, n, chunk):
                hi = min(n, lo + chunk)
                wd = None if widths is None else widths[lo:hi]
                q.put((lo, model.beam_frontend(images[lo:hi], k=k, widths=wd, want_candidates=codec.skip_search)))
        except BaseException as exc:      # surfaced in the consumer thread
            err.append(exc)
        finally:
            q.put(None)

    t = threading.Thread(target=producer, daemon=True)
    t.start()
    out = [None] * n
    while True:
        item = q.get()
        if item is None:
            break
        lo, fe = item
        for i, text in enumerate(codec.decode_frontend(fe)):
            out[lo + i] = text
    t.join()
    if err:
        raise err[0]
    return out
