# Round-4 profile set, one GPU call (run from the repo root on the GPU box: bash tools/collect_profiles.sh).  Counter passes carry
# --pmc only (no trace domains); the program goes directly after `--`.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && R=r04 && \
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}k -o k -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fresh > gpurun_out/${R}k.log 2>&1 && echo stats-done && \
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fresh > gpurun_out/pmc_f.log 2>&1 && echo fetch-done && \
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fresh > gpurun_out/pmc_w.log 2>&1 && echo write-done && \
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d gpurun_out/pmc_b -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fresh > gpurun_out/pmc_b.log 2>&1 && echo busy-done && \
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_l -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fresh > gpurun_out/pmc_l.log 2>&1 && echo lds-done && \
python3 tools/pmc_traffic.py traffic gpurun_out/pmc_f/p_counter_collection.csv gpurun_out/pmc_w/p_counter_collection.csv 4194304 > gpurun_out/${R}_pmc_traffic.json && \
python3 tools/pmc_traffic.py busy gpurun_out/pmc_b/p_counter_collection.csv,gpurun_out/pmc_l/p_counter_collection.csv 4194304 > gpurun_out/${R}_pmc_mfma.json && echo json-done && rm -rf gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_b gpurun_out/pmc_l && \
python3 bench.py --steps 10 --warmup 3 > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err && echo bench-done && \
python3 bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${R}_bench_cfg3.json 2> gpurun_out/${R}_bench_cfg3.err && \
python3 bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${R}_bench_cfg5.json 2> gpurun_out/${R}_bench_cfg5.err && echo cfg-done
