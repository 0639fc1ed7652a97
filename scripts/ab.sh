#!/bin/bash
# ONE parametrised A/B runner (replaces the twelve scripts/ab_*.sh of rounds 1 - 3): interleaved rounds of several ARMS of
# one bring-up benchmark on one device (cdna_hip_programming.md rule 24: never rank builds across processes / devices).
#
#   scripts/ab.sh <search|gemm|attention|encode> [-r ROUNDS] [-g GREP] -a ARM [-a ARM ...] [-- bench arguments]
#
# ARM = name[:lib][:ENV=VAL,ENV=VAL...]
#   lib   ""          the production library
#         ablation    libimagescry_hip_ablation.so            (python -m imagescry_amd.build --ablation)
#         <variant>   libimagescry_hip_<variant>.so           (python -m imagescry_amd.build --variant=<variant> -D...)
#   ENV   the ISC_* switches an ablation / variant build reads (ISC_DEBUG_MODE, ISC_NO_NT, ISC_SEG_TILES, ISC_GEMM_DEBUG, ...)
#
# The experiments of LABLOG.md as invocations (what used to be a script each):
#   ab_nt          ab.sh search -r 3 -a nt:ablation -a default:ablation:ISC_NO_NT=1 -- 1250000x16
#   ab_seg         ab.sh search -a s100000:ablation:ISC_SEG_TILES=100000 -a s128:ablation:ISC_SEG_TILES=128 ... -- 10000000x1024
#   ab_first_ratio ab.sh search -a off:ablation:ISC_FIRST_RATIO=1048576 -a r16:ablation:ISC_FIRST_RATIO=16 -- 10000000x256
#   ab_headline    ab.sh search -r 3 -a base:ablation -a noprio:ablation:ISC_NO_STATIC_PRIO=1 -a split:ablation:ISC_FORCE_SPLIT=1 -- 10000000x1024
#   ab_noq         ab.sh search -a m12:ablation:ISC_DEBUG_MODE=12 -a m41:ablation:ISC_DEBUG_MODE=41 -a m17:ablation:ISC_DEBUG_MODE=17 -- 10000000x1024
#   ab_asym        ab.sh search -a sym:sym -a a6:a6 -a a5:a5 -- 10000000x1024      (variants: -DISC_ASYM_MBLO=<n>)
#   ab_hm          ab.sh search -a hm:hm0:ISC_DEBUG_MODE=46 -a base:hm0:ISC_DEBUG_MODE=12 -- 10000000x1024
#   ab_gemm        ab.sh gemm -r 1 -a d0:ablation:ISC_GEMM_DEBUG=0 -a d1:ablation:ISC_GEMM_DEBUG=1 ...
#   ab_gemm_split  ab.sh gemm -a split:ablation -a nosplit:ablation:ISC_GEMM_NO_SPLIT=1
#   ab_gemm_seg    ab.sh gemm -r 1 -a seg0:ablation:ISC_GEMM_SEG=0 -a seg4:ablation:ISC_GEMM_SEG=4
#   ab_gemm_phase  ab.sh gemm -g "proj|fc2" -a off:ablation:ISC_GEMM_PHASE=-1 -a p2g4:ablation:ISC_GEMM_PHASE=2,ISC_GEMM_PHASE_GROUPS=4
#   ab_gemm_spread ab.sh gemm -r 3 -a gs80:gs80 -a gs22:gs22 -a gs32:gs32      (variants: -DISC_GS_NI=<n> -DISC_GS_NH=<n>)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
bench=$1; shift
case "$bench" in
  search) script=scripts/quick_search_bench.py ;;
  gemm) script=scripts/quick_gemm_bench.py ;;
  attention) script=scripts/quick_attention_bench.py ;;
  encode) script=scripts/quick_encode_bench.py ;;
  *) echo "usage: $0 <search|gemm|attention|encode> [-r rounds] [-g grep] -a arm ... [-- args]"; exit 2 ;;
esac
rounds=2; pattern="."; arms=()
while [ $# -gt 0 ]; do
  case "$1" in
    -r) rounds=$2; shift 2 ;;
    -g) pattern=$2; shift 2 ;;
    -a) arms+=("$2"); shift 2 ;;
    --) shift; break ;;
    *) break ;;
  esac
done
[ ${#arms[@]} -gt 0 ] || { echo "no arms (-a name[:lib][:ENV=VAL,...])"; exit 2; }
for round in $(seq 1 "$rounds"); do
  for arm in "${arms[@]}"; do
    IFS=':' read -r name lib envs <<< "$arm"
    (
      if [ -n "$lib" ]; then
        export ISC_ALLOW_ABLATION=1 ISC_LIB="$PWD/imagescry_amd/libimagescry_hip_$lib.so"
      fi
      IFS=',' read -ra kv <<< "$envs"
      for e in "${kv[@]}"; do [ -n "$e" ] && export "$e"; done
      python3 "$script" "$@" 2>&1 | grep -v amdgpu.ids | grep -E "$pattern" | sed "s/^/[$name r$round] /"
    )
  done
done
