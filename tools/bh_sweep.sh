cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for nb in 512 1024 1536 1792 2048; do echo "BLOCKS=$nb"; NB_BH_BLOCKS=$nb timeout -k 10 120 python tools/bench_bh.py 0.5 16777216 2>&1 | cut -c1-160; done > gpurun_out/bh_blocks.txt 2>&1
cat gpurun_out/bh_blocks.txt
