cd /tmp && export TMPDIR=/tmp
for dbg in 0 1 2 3 7; do
  rm -rf /tmp/hp
  FC_DEBUG_RG=$dbg SHAPE=net WHICH=b rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hp -o h -- python3 "$GRAFT_REPO_ROOT/tools/time_head.py" > /dev/null 2>&1
  python3 - /tmp/hp/h_kernel_stats.csv $dbg <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'fc_rgemm_kernel' in r['Name']:
        print(f"dbg {sys.argv[2]}: rgemm avg {float(r['AverageNs']) / 1e3:7.1f} us")
PY
done
