"""GPU box: device time per frame and per stage of a named scene (mr_render, one stream, frame-only mode)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import scenes
api = scenes.product_api()
for name in sys.argv[1:]:
    sc = scenes.build(api, name)
    be = sc._backend()
    shadows = name not in scenes.NO_SHADOW
    be.render(sc, shadows=shadows)                    # counted frame: the fragment counters
    st = be.last_stats
    for _ in range(12):                               # timed the way Scene.render() renders (frame only)
        be.render(sc, shadows=shadows, counters=False)
    kt, n = be.kernel_times(10)
    frags = 2 * st["frag_tri"] + st["frag_quad"]
    print(name, "frame_ms", round(kt["frame"], 4), "Mfrag/s", round(frags / kt["frame"] / 1e3, 1), "fragments", frags,
          {k: round(v * 1e3, 1) for k, v in kt.items()}, "copy_ms", round(be.last_stats["gpu_ms_copy"], 3), flush=True)
    sc.close()
