import sys
sys.path.insert(0, '.')
from disco_amd import buildgraph
g = buildgraph.BuildGraph(min_overlap=40, device=0)
print("copy GB/s", [round(g.measure_hbm(b << 20, 5)) for b in (256, 1024, 4096)])
print("gather GB/s (table MB -> GB/s)", {mb: round(g.measure_gather(mb << 20, 3)) for mb in (64, 512, 3200, 12800)})
