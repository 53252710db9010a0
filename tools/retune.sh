#!/bin/bash
# In-forward retune on a GPU box: sweep -> tile_table.inc -> rebuild -> A/B against the cost model.  Outputs under gpurun_out/.
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 300 python tools/fwd_tune.py 12 1 3 > gpurun_out/fwd_tune.log 2>&1 || { tail -20 gpurun_out/fwd_tune.log; exit 1; }
cat gpurun_out/fwd_tune.log
cp pnpinversion_amd/csrc/tile_table.inc gpurun_out/tile_table_prev.inc
python tools/gen_tile_table.py pnpinversion_amd/csrc/tile_table.inc gpurun_out/fwd_tune_b12.json gpurun_out/fwd_tune_b1.json gpurun_out/fwd_tune_b3.json
cp pnpinversion_amd/csrc/tile_table.inc gpurun_out/tile_table_new.inc
timeout 400 python -m pnpinversion_amd.build > gpurun_out/rebuild.log 2>&1 || { tail -20 gpurun_out/rebuild.log; exit 1; }
timeout 200 python tools/fwd_ab.py 1 12 3 > gpurun_out/fwd_ab_r2f.log 2>&1; cat gpurun_out/fwd_ab_r2f.log
timeout 300 python -m pytest tests/test_gpu_loops.py -x -q -k "reconstruction or pruned" 2>&1 | tail -5
timeout 300 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_r2f.json 2> gpurun_out/bench_r2f.err; cat gpurun_out/bench_r2f.json
