# LDS bank-conflict counters of the struct-stage kernels alone (tools/bench_stage.py), this build vs tools/bin/libmgvae_base.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in new base; do
  if [ $v = base ]; then export MGV_LIB=$GRAFT_REPO_ROOT/tools/bin/libmgvae_base.so; else unset MGV_LIB; fi
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_ls_$v -o p -- python3 tools/bench_stage.py 64 2 > gpurun_out/pmc_ls_$v.log 2>&1
  python3 - $v <<'PY'
import csv, collections, sys
v = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for r in csv.DictReader(open('gpurun_out/pmc_ls_%s/p_counter_collection.csv' % v)):
    acc[r['Kernel_Name']][r['Counter_Name']][r.get('Dispatch_Id', '0')] += float(r['Counter_Value'])
for k in sorted(acc):
    if 'struct_stage' not in k or 'reduce' in k:
        continue
    c = acc[k]
    # largest launches only
    ref = c['SQ_BUSY_CU_CYCLES']
    mx = max(ref.values())
    ids = [d for d, x in ref.items() if x >= 0.5 * mx]
    tot = lambda n: sum(c[n][d] for d in ids)
    print('%s %-40s launches %d: lds_active/cu %.3f  bank_conflict/lds_active %.3f  bank_conflict/cu %.3f  wait_inst_lds/wave %.4f  insts_lds/launch %.3g' % (
        v, k.split('(')[0][-40:], len(ids), tot('SQ_LDS_IDX_ACTIVE') / tot('SQ_BUSY_CU_CYCLES'), tot('SQ_LDS_BANK_CONFLICT') / tot('SQ_LDS_IDX_ACTIVE'),
        tot('SQ_LDS_BANK_CONFLICT') / tot('SQ_BUSY_CU_CYCLES'), tot('SQ_WAIT_INST_LDS') / tot('SQ_WAVE_CYCLES'), tot('SQ_INSTS_LDS') / len(ids)))
PY
  rm -rf gpurun_out/pmc_ls_$v
done
