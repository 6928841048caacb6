cd $GRAFT_REPO_ROOT
bash tools/refresh_profiles.sh r04 > gpurun_out/refresh_r04.log 2>&1
bash tools/pmc_pass.sh r04 FETCH_SIZE >> gpurun_out/refresh_r04.log 2>&1
bash tools/pmc_pass.sh r04 WRITE_SIZE >> gpurun_out/refresh_r04.log 2>&1
bash tools/pmc_pass.sh r04 TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_ATOMIC_sum >> gpurun_out/refresh_r04.log 2>&1
bash tools/pmc_pass.sh r04 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM >> gpurun_out/refresh_r04.log 2>&1
tail -5 gpurun_out/refresh_r04.log; ls -la gpurun_out | grep r04
