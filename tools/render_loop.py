"""Render a scene N times through mr_render (for rocprofv3 runs); a third argument "frame-only"
renders the way Scene.render() and bench.py do (no MR_FRAME_COUNTERS)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import scenes
name = sys.argv[1] if len(sys.argv) > 1 else "c4_torus200k_1080p"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
api = scenes.product_api()
sc = scenes.build(api, name)
be = sc._backend()
counters = not (len(sys.argv) > 3 and sys.argv[3] == "frame-only")
for _ in range(n):
    be.render(sc, shadows=name not in scenes.NO_SHADOW, counters=counters)
print(be.last_stats)
