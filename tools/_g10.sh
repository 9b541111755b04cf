bash tools/profile_all.sh r03g > gpurun_out/r03g_profile_all.log 2>&1
tail -3 gpurun_out/r03g_profile_all.log
