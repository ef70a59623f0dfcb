cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && export NB_NO_GRAPH=1 NB_SR_TILES=1 NB_SR_W=8
O=gpurun_out/prof_farm_tiles; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 tools/farm_job_profile.py 400 > $O/out.txt 2>&1
python3 tools/summarize_kernel_stats.py $O/t 22
