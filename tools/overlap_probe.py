"""does co-scheduling phases of DIFFERENT kinds pay? Two independent contexts of N reads each on one GPU, passes driven by two host
threads with a phase shift, against the same passes one after the other.   python tools/overlap_probe.py [N=25000000] [PASSES=6]
(DISCO_PROBE_WAVES / DISCO_VERIFY_WAVES / DISCO_TR_WAVES limit the resident workgroups per CU of the persistent kernels.)"""
import sys, threading, time
sys.path.insert(0, '.')
from disco_amd import buildgraph, readgen

n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
gs = []
for s in (1, 2):
    g = buildgraph.BuildGraph(min_overlap=40)
    g.generate_reads(readgen.GenSpec.coverage(40 + s, n, 150, 30.0, n_contigs=max(1, n * 5 // 5_000_000)))
    g.run_graph(); g.synchronize()
    gs.append(g)


def passes(g, k, delay=0.0):
    time.sleep(delay)
    for _ in range(k):
        g.run_graph()
    g.synchronize()


t0 = time.perf_counter(); passes(gs[0], K); t_one = (time.perf_counter() - t0) / K
t0 = time.perf_counter(); passes(gs[0], K); passes(gs[1], K); t_seq = (time.perf_counter() - t0) / (2 * K)
for delay in (0.0, t_one * 0.5):
    th = [threading.Thread(target=passes, args=(gs[i], K, delay * i)) for i in range(2)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    print(f"two streams, shift {delay * 1e3:.0f} ms: {(time.perf_counter() - t0 - delay) / (2 * K) * 1e3:.1f} ms per pass")
print(f"one after the other: {t_seq * 1e3:.1f} ms per pass (alone: {t_one * 1e3:.1f})")
