"""End-to-end files -> text throughput of the drop-in CLI on the GPU box (host decode + device resize + forward +
greedy decode + CER bookkeeping):  python tools/gpu_e2e_files.py [n_files] [batch] [workers] [ragged]
Writes n synthetic half-height PNG line images (64 x 1000 -> resized on the device to 128 x 2000; with "ragged" the
source widths vary between 400 and 800, so that every batch pads to a different width <= 1600) and a
test_img_id_gt.txt into a temporary folder, runs `test.py -bm` on it and reports lines/s from its own clock."""
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hctr_amd  # noqa: E402
from PIL import Image  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
workers = int(sys.argv[3]) if len(sys.argv) > 3 else 16
ragged = len(sys.argv) > 4 and sys.argv[4] == "ragged"
synth = hctr_amd.synth
with tempfile.TemporaryDirectory() as d:
    os.makedirs(os.path.join(d, "test"))
    imgs = synth.make_line_images(n, 2000, seed=77)
    with open(os.path.join(d, "test_img_id_gt.txt"), "w", encoding="utf-8") as f:
        for i in range(n):
            half = imgs[i][::2, ::2]
            if ragged:
                half = half[:, :400 + int(synth.uniform01(77, 5, 1, offset=i)[0] * 400)]
            Image.fromarray(half).save(os.path.join(d, "test", "%06d.png" % i))
            f.write("%06d.png,%s\n" % (i, synth.characters()[i % 100]))
    for w in (0, workers):
        t0 = time.time()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "test.py"), "-m", "hctr", "-f", "synthetic", "-i", d, "-bm",
                            "-b", str(batch), "-dm", "greedy-search", "-jw", str(w), "-pf", "1000"],
                           capture_output=True, text=True)
        wall = time.time() - t0
        m = re.search(r"Total Test CER: \S+ \(([0-9.]+)s\)", r.stdout)
        if r.returncode != 0 or not m:
            print(r.stdout[-1000:], r.stderr[-2000:])
            raise SystemExit(1)
        loop = float(m.group(1))
        print("workers=%2d: %d files in %.2f s inside the loop -> %.1f lines/s (process wall %.1f s incl. start-up)" %
              (w, n, loop, n / loop, wall))
