# A/B of two builds of the library on one box: tools/ab_old/libternary_hip.so against the tree's own.
# usage: bash tools/ab_run.sh <out.jsonl> "M,N,K ..." [plan ...]
OUT=$1; SHAPES=$2; shift 2
PLANS=${@:-auto}
rm -rf /tmp/pkg_old && mkdir -p /tmp/pkg_old && cp -r ternary-vllm_amd /tmp/pkg_old/ && cp ternary_vllm_amd.py /tmp/pkg_old/ 2>/dev/null
cp -r ternary_vllm_amd /tmp/pkg_old/ 2>/dev/null
cp tools/ab_old/libternary_hip.so /tmp/pkg_old/ternary-vllm_amd/libternary_hip.so || exit 1
: > $OUT
for rep in 1 2; do
  for pl in $PLANS; do
    if [ "$pl" = auto ]; then unset TVL_FORCE_PLAN; else export TVL_FORCE_PLAN=$pl; fi
    TVL_PKG_ROOT=/tmp/pkg_old timeout -k 10 300 python tools/time_plans.py $SHAPES 2>/dev/null | sed "s/\"plan\": \"\([^\"]*\)\"/\"plan\": \"old-\1\"/" >> $OUT || exit 1
    timeout -k 10 300 python tools/time_plans.py $SHAPES 2>/dev/null | sed "s/\"plan\": \"\([^\"]*\)\"/\"plan\": \"new-\1\"/" >> $OUT || exit 1
  done
done
python - $OUT <<'PY'
import json,collections,sys
d=collections.defaultdict(lambda: collections.defaultdict(list))
for l in open(sys.argv[1]):
    r=json.loads(l); d[(r['m'],r['n'],r['k'])][r['plan']].append(round(r['us'],2))
for s,v in d.items():
    ks=sorted(set(k[4:] for k in v))
    print(s, ' | '.join(f"{k}: old {min(v['old-'+k])} new {min(v['new-'+k])} ({100*(min(v['new-'+k])/min(v['old-'+k])-1):+.1f}%)" for k in ks))
PY
