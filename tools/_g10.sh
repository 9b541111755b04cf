bash tools/profile_all.sh r03f > gpurun_out/r03f_profile_all.log 2>&1
tail -5 gpurun_out/r03f_profile_all.log
