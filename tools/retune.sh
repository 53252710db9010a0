#!/bin/bash
# In-forward retune on a GPU box: sweep -> tile_table.inc -> rebuild -> A/B against the cost model.  Outputs under gpurun_out/.
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
ROWS="${RETUNE_ROWS:-12 1 3 96 8 4}"
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q 2>&1 | tail -4
timeout 500 python tools/fwd_tune.py $ROWS > gpurun_out/fwd_tune.log 2>&1 || { tail -20 gpurun_out/fwd_tune.log; exit 1; }
grep -v "^  (" gpurun_out/fwd_tune.log
cp pnpinversion_amd/csrc/tile_table.inc gpurun_out/tile_table_prev.inc
FWD_TUNE_WORK=vae timeout 300 python tools/fwd_tune.py 1 2 >> gpurun_out/fwd_tune.log 2>&1        # VAE encode + decode at 512 x 512
FWD_TUNE_WORK=ctxgrad timeout 300 python tools/fwd_tune.py 1 >> gpurun_out/fwd_tune.log 2>&1     # null-text iteration (dgrad shapes of the reverse walk)
# the headline schedule's row counts first (a shape two sweeps share keeps the first file's measurement), then the VAE, then the walk
python tools/gen_tile_table.py pnpinversion_amd/csrc/tile_table.inc $(for r in $ROWS; do echo gpurun_out/fwd_tune_b$r.json; done) gpurun_out/fwd_tune_b1vae.json gpurun_out/fwd_tune_b2vae.json gpurun_out/fwd_tune_b1bwd.json
cp pnpinversion_amd/csrc/tile_table.inc gpurun_out/tile_table_new.inc
timeout 400 python -m pnpinversion_amd.build > gpurun_out/rebuild.log 2>&1 || { tail -20 gpurun_out/rebuild.log; exit 1; }
FWD_AB_ARMS=default,no_table timeout 200 python tools/fwd_ab.py 1 12 3 > gpurun_out/fwd_ab_retuned.log 2>&1; cat gpurun_out/fwd_ab_retuned.log
timeout 300 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_retuned.json 2> gpurun_out/bench_retuned.err; cat gpurun_out/bench_retuned.json
