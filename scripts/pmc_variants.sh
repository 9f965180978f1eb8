set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/pmc3 -o p -- python scripts/fused_stats.py 100 --stats > gpurun_out/pmc3.log 2>&1
python - <<'PY'
import csv, collections
rows=list(csv.DictReader(open('gpurun_out/pmc3/p_counter_collection.csv')))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in rows:
    k=r['Kernel_Name'].split('(')[0]
    if 'k_fused' in k:
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
        if r['Counter_Name']=='SQ_INSTS_VALU': n[k]+=1
for k in agg:
    print(k, n[k], {c: round(v/n[k]/1e9,3) for c,v in agg[k].items()})
PY
