cd $GRAFT_REPO_ROOT
bash tools/profile_r03.sh > gpurun_out/s2_prof.log 2>&1; echo "rc=$?" >> gpurun_out/s2_prof.log
tail -30 gpurun_out/s2_prof.log
