"""GPU box: the frame counters of a named scene (mr_get_stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import scenes
api = scenes.product_api()
for name in sys.argv[1:]:
    sc = scenes.build(api, name)
    be = sc._backend()
    be.render(sc, shadows=name not in scenes.NO_SHADOW)
    print(name, be.last_stats, flush=True)
    sc.close()
