cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02k -o k -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02k.log 2>&1 && echo stats-done && \
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_f.log 2>&1 && echo fetch-done && \
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_w.log 2>&1 && echo write-done && \
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_b -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_b.log 2>&1 && echo busy-done && \
ls gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_b && \
python3 tools/pmc_traffic.py traffic gpurun_out/pmc_f/p_counter_collection.csv gpurun_out/pmc_w/p_counter_collection.csv 4194304 > gpurun_out/r02_pmc_traffic.json && \
python3 tools/pmc_traffic.py busy gpurun_out/pmc_b/p_counter_collection.csv 4194304 > gpurun_out/r02_pmc_mfma.json && echo json-done && rm -rf gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_b
