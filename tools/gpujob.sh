#!/bin/bash
# One parametrised GPU-box job script (replaces the per-call run_rNN*.sh scripts of earlier rounds).
#   gpurun --timeout 1500 -- 'bash tools/gpujob.sh <outdir> <step> [<step> ...]'
# Steps (each writes under gpurun_out/<outdir>/, prints a short tail):
#   tests[:<pytest -k expr>]      pytest -m gpu (whole suite, or the subset)
#   bench[:ENV=V,ENV=V]           one bench.py line (no cpu baseline / other configs) under the given switches
#   ab:SETA/SETB[/SETC][@runs]    tools/gpu_ab_wall.py with comma-separated switch sets (use X=0 for the default)
#   convbench[:ENV=V,...]         tools/gpu_conv_bench.py (cold operands, per layer)
#   wgradbench[:ENV=V,...]        tools/gpu_wgrad_bench.py + tools/gpu_wgrad3_bench.py
#   dwprobe[:ENV=V,...]           tools/gpu_dw_probe.py
#   naprobe[:ENV=V,...]           tools/gpu_na_probe.py
#   serial[:ENV=V,...]            tools/gpu_prof_step.py --serial (per-kernel table, kernels alone)
#   timeline                      tools/gpu_step_timeline.py
#   profiles                      bash tools/collect_profiles.sh <outdir>_prof (bench line + rocprofv3 stats + PMC passes)
#   py:<script>[:args]            python tools/<script> args
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
D=$1; shift
O=$R/gpurun_out/$D; mkdir -p $O
envrun() {  # envrun "A=1,B=2" cmd...
  local sw=$1; shift
  if [ -n "$sw" ]; then env $(echo "$sw" | tr ',' ' ') "$@"; else "$@"; fi
}
n=0
for step in "$@"; do
  n=$((n+1))
  kind=${step%%:*}; arg=""; [ "$step" != "$kind" ] && arg=${step#*:}
  tag=$(printf "%02d_%s" $n $kind)
  echo "=== $step"
  case $kind in
    tests)
      if [ -n "$arg" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -k "$arg" > $O/$tag.log 2>&1
      else timeout 1800 python -m pytest tests -m gpu -x -q > $O/$tag.log 2>&1; fi
      echo "pytest rc $?" | tee -a $O/$tag.log; tail -4 $O/$tag.log ;;
    bench)
      envrun "$arg" timeout 400 python bench.py --no-cpu-baseline --no-other-configs > $O/$tag.json 2> $O/$tag.err
      tail -1 $O/$tag.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%s  %.3f ms  %.1f img/s' % ('$arg', d['ms_per_step'], d['value']))" ;;
    ab)
      runs=3; sets=$arg
      case $arg in *@*) runs=${arg##*@}; sets=${arg%@*};; esac
      IFS='/' read -ra S <<< "$sets"
      timeout 2400 python tools/gpu_ab_wall.py "${S[@]}" --runs $runs 2>&1 | tee $O/$tag.log ;;
    convbench) envrun "$arg" timeout 600 python tools/gpu_conv_bench.py > $O/$tag.log 2>&1; cat $O/$tag.log ;;
    wgradbench) envrun "$arg" timeout 600 python tools/gpu_wgrad_bench.py > $O/$tag.log 2>&1; envrun "$arg" timeout 600 python tools/gpu_wgrad3_bench.py >> $O/$tag.log 2>&1; cat $O/$tag.log ;;
    dwprobe) envrun "$arg" timeout 600 python tools/gpu_dw_probe.py > $O/$tag.log 2>&1; tail -40 $O/$tag.log ;;
    naprobe) envrun "$arg" timeout 600 python tools/gpu_na_probe.py > $O/$tag.log 2>&1; tail -40 $O/$tag.log ;;
    serial) envrun "$arg" timeout 900 python tools/gpu_prof_step.py --serial --top 200 > $O/$tag.log 2>&1; head -70 $O/$tag.log ;;
    timeline) timeout 900 python tools/gpu_step_timeline.py > $O/$tag.log 2>&1; tail -60 $O/$tag.log ;;
    profiles) bash tools/collect_profiles.sh ${D}_prof ;;
    py)
      script=${arg%%:*}; pargs=""; [ "$arg" != "$script" ] && pargs=${arg#*:}
      timeout 1200 python tools/$script $pargs > $O/$tag.log 2>&1; echo "rc $?"; tail -60 $O/$tag.log ;;
    *) echo "unknown step $step" ;;
  esac
done
