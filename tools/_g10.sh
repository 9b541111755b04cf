bash tools/profile_all.sh r03d > gpurun_out/r03d_profile_all.log 2>&1
tail -5 gpurun_out/r03d_profile_all.log
ls gpurun_out | grep r03d | head -80
