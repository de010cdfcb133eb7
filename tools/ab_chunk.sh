cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab_chunk
timeout -k 10 500 python tools/ab_run.py tools/ab_early.py > gpurun_out/ab_chunk/early.txt 2>&1
for v in p2 p1 p0 p2 p0; do
  echo "== $v" >> gpurun_out/ab_chunk/configs.txt
  QBP_LIB_PATH=$GRAFT_REPO_ROOT/build/variants/libqbp_$v.so timeout -k 10 200 python tools/bench_configs.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print(d['case'], 'early %.4g (%.3f ms)  forced %.4g' % (d['early_exit_syn_per_s'], d['early_exit_ms'], d['forced_50_syn_per_s']))
" >> gpurun_out/ab_chunk/configs.txt
done
cat gpurun_out/ab_chunk/early.txt gpurun_out/ab_chunk/configs.txt
