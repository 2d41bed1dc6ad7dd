cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for sh in ref ext ext_noshadow ext_noplane ext_nospec; do
  for v in auto P4; do
    rm -rf /tmp/ip; timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d /tmp/ip -- python3 $R/tools/insts_probe.py c2_1920x1080_n64 $sh $v > /tmp/ip.log 2>&1 || { echo fail; tail -3 /tmp/ip.log; exit 1; }
    python3 - "$sh" "$v" <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(list)
for f in glob.glob('/tmp/ip/*/*_counter_collection.csv'):
    per=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if 'rc_draw' in r['Kernel_Name']: per[r['Dispatch_Id']][r['Counter_Name']]+=float(r['Counter_Value'])
    for d,c in per.items():
        for k,v in c.items(): acc[k].append(v)
print(sys.argv[1], sys.argv[2], {k: round(sum(v)/len(v)) for k,v in acc.items()})
PY
  done
done
