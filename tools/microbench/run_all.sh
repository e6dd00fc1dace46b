#!/bin/bash
# The micro-benchmarks DESIGN section 8's leads want answered before any kernel is touched (a minute of GPU in all):
#   occupancy_probe    do two 512-lane workgroups with 80 KB of LDS still share a CU (k_lookup with end masks + a candidate list)?
#   latency_probe      what does a dependent plain / device-scope load / CAS cost from L2, the Infinity Cache, HBM -- idle and loaded?
#   entry_key_probe    the claims with the claimant's key inside the slot entry (one read) against today's two reads
#   random_lines_probe (round 4) how many random 16-byte reads per second the L2 / Infinity Cache / HBM serve: the roofline of k_lookup
#   claims_probe       (round 3) device-scope loads / CAS under a Zipf law, staleness of plain loads across XCDs
# usage (GPU box, repo root): tools/microbench/run_all.sh [outdir under gpurun_out/]
cd "$(dirname "$0")"
for f in occupancy_probe latency_probe entry_key_probe claims_probe random_lines_probe; do
  [ -x $f ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $f $f.hip || exit 1
done
out=../../gpurun_out/${1:-microbench}; mkdir -p "$out"
for f in occupancy_probe entry_key_probe claims_probe latency_probe random_lines_probe; do
  echo "== $f"; timeout 120 ./$f 2>&1 | tee "$out/$f.txt"
done
