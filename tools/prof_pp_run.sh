cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && export NB_NO_GRAPH=1
O=gpurun_out/prof_pp; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 tools/run_pp_reference.py 40 64 1 f32 2 pp > $O/out.txt 2>&1
python3 tools/summarize_kernel_stats.py $O/t 10
