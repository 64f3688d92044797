import sys, faulthandler
sys.path.insert(0, '.')
from tests import golden_util as gu
from disco_amd import buildgraph
name = sys.argv[1] if len(sys.argv) > 1 else "repeats_8k"
reads, fidx, mo = gu.case_inputs(name)
g = buildgraph.BuildGraph(min_overlap=mo)
g.upload_ascii(reads)
for step in ("build_index", "probe", "mark_contained", "build_edges", "transitive_reduce"):
    print("->", step, flush=True)
    r = getattr(g, step)()
    g.synchronize()
    print("   ok", r, flush=True)
print(g.counters())
