cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03b/agcn_clip -- python3 $R/tools/agcn_prof.py 64 6 > $R/gpurun_out/prof_r03b/agcn_clip.log 2>&1
grep AGCN_PASS $R/gpurun_out/prof_r03b/agcn_clip.log
