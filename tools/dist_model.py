#!/usr/bin/env python3
"""Time model of an 8-rank pass from the measured profile of 8 ranks serialised on ONE GPU (tools/dist_profile.py + profiles/prof_dist.sh):
   python tools/dist_model.py profiles/r05_dist8.json [profiles/r05_dist8_kernels.json] [LINK_GBS=50]
NOT a measurement on 8 GPUs: kernels = the job's kernel time / ranks (the work is balanced to 0.2 %: own_reads_max_over_mean);
every exchange = the bytes a rank sends to ONE peer / the link rate (every peer pair has a link of its own on the xGMI mesh, so a
rank's 7 blocks travel at once); a host wait = 20 us, an operation on a communicator = 10 us of launch."""
import json
import sys

prof = json.load(open(sys.argv[1]))
ktab = json.load(open(sys.argv[2] if len(sys.argv) > 2 else sys.argv[1].replace(".json", "_kernels.json")))
link = float(sys.argv[3]) if len(sys.argv) > 3 else 50.0
G = prof["ranks"]
k = ktab["kernels"]
single = ktab["single_gpu_ms"]
total = ktab["sum_over_ranks_ms"]
# kernels of the SECOND stream (behind keys .. probe / behind selection .. emission): the packing of the gathered reads, and the reduction of the
# containment keys (loop_min_i64_kernel: the in-process transport's stand-in for the reduction RCCL does in its own kernels)
bulk = sum(v["ranks_ms"] for n, v in k.items() if n.startswith(("unpack_rows_kernel", "pack_rows_kernel", "unpack_lens_kernel", "loop_min_i64_kernel")))
b = prof["bytes_sent_per_rank_mean"]


def ms(nbytes):  # a rank's bytes to its G - 1 peers, one link each
    return nbytes / (G - 1) / (link * 1e9) * 1e3


main_kernels = (total - bulk) / G
terms = {"keys (all-gather)": ms(b.get("keys", 0)), "own rows (all-to-all)": ms(b.get("reads_dealt", 0)), "index records (all-to-all)": ms(b["index_records"]),
         "index slices (all-gather)": ms(b["index_shards"]), "containment (bitmaps; keys too when they are on the path)": ms(b["contain"]), "row requests + degrees": ms(b["row_requests"]),
         "row data": ms(b["row_data"]), "survivor push": ms(b["push"])}
hidden = {"all reads (all-gather, second communicator: behind keys .. probe)": ms(b["reads"]), "its unpacking (second stream)": bulk / G,
          "containment keys (reduce-scatter, second communicator: behind selection .. emission)": ms(b.get("contain_keys", 0))}
host = prof["host_syncs_per_pass"] * 0.020 + prof["comm_ops_per_pass"] * 0.010
exposed = sum(terms.values())
t = main_kernels + exposed + host
print(f"placement {prof.get('placement', 'id ranges')}; work inflation {total / single:.3f} ({total:.1f} ms over the ranks / {single:.1f} ms on one GPU)")
print(f"kernels on the pass's stream  {main_kernels:6.2f} ms  (= ({total:.1f} - {bulk:.1f} on the second stream) / {G})")
for n, v in terms.items():
    print(f"  {n:32s} {v:6.2f} ms")
print(f"exchanges nothing hides        {exposed:6.2f} ms at {link:.0f} GB/s per link and direction")
for n, v in hidden.items():
    print(f"  (hidden) {n:70s} {v:6.2f} ms")
print(f"host: {prof['host_syncs_per_pass']} waits, {prof['comm_ops_per_pass']} operations  {host:6.2f} ms")
print(f"pass at G = {G}                  {t:6.2f} ms  ->  {single / t:.2f} x of the single-GPU {single:.1f} ms")
