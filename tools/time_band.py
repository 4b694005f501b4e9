"""GPU box: device time per stage when this device renders only one band of H/N rows (what a rank of an
N-GPU run does per frame, minus the all-gather).  usage: tools/time_band.py <scene> <N> [band index]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import scenes
api = scenes.product_api()
name, n = sys.argv[1], int(sys.argv[2])
k = int(sys.argv[3]) if len(sys.argv) > 3 else n // 2
sc = scenes.build(api, name)
be = sc._backend()
h = sc.resolution[0]
rows = h // n
for _ in range(12):
    be.render(sc, shadows=True, row_band=(k * rows, (k + 1) * rows), counters=False)
kt, _ = be.kernel_times(10)
print(name, f"band {k}/{n}", {a: round(b * 1e3, 1) for a, b in kt.items()}, flush=True)
