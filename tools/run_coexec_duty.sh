# VALU / MFMA co-execution at partial matrix duty: wall times, then the SQ counters per case (binary directly after `--`).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
tools/bin/coexec_duty 4000 > gpurun_out/r04_coexec_duty.txt 2>&1 && \
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_cd -o p -- tools/bin/coexec_duty 1000 > gpurun_out/pmc_cd.log 2>&1 && \
python3 - <<'PY' >> gpurun_out/r04_coexec_duty.txt
import csv, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open('gpurun_out/pmc_cd/p_counter_collection.csv')):
    acc[r['Kernel_Name']][r['Counter_Name']].append((int(r.get('Dispatch_Id', 0)), float(r['Counter_Value'])))
print('\nSQ counters per case (second, timed launch of each kernel; rocprofv3 --pmc, counters only):')
print('%-44s %10s %10s %10s %10s' % ('kernel k_duty<GAP, KIND(0 pk_fma,1 fma,2 exp), WHO(1 mfma,2 valu,3 both)>', 'mfma_busy', 'valu_busy', 'coexec', 'coexec/mfma'))
for k in sorted(acc):
    m = re.search(r'k_duty<(\d+), (\d+), (\d+)>', k)
    if not m:
        continue
    last = max(d for d, _ in acc[k]['SQ_BUSY_CU_CYCLES'])
    v = {c: sum(x for d, x in vals if d == last) for c, vals in acc[k].items()}
    cu = v['SQ_BUSY_CU_CYCLES']
    mb, vb, co = v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4 * cu), v.get('SQ_ACTIVE_INST_VALU', 0) / cu, v.get('SQ_VALU_MFMA_COEXEC_CYCLES', 0) / (4 * cu)
    print('k_duty<%3s, %s, %s> %34s %10.3f %10.3f %10.3f %10.3f' % (m.group(1), m.group(2), m.group(3), '', mb, vb, co, co / mb if mb else 0.0))
PY
rm -rf gpurun_out/pmc_cd; echo coexec-done
