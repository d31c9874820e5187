#!/bin/bash
# Same-box A/B of WHOLE ROUNDS: each round's own tree (its bench.py, its host mirror, its library) against this HEAD, run
# interleaved inside one gpurun call, so that "round N = round N-1 +- x %" is a same-box reading and not a comparison of two
# boxes of the pool (which differ by +-4 % in the accumulation kernels).
#
#   here (no GPU):    tools/ab_binaries.sh prepare r03=23254aa r04=86542de      # git archive -> build_ab/<tag>/, cross-compiled
#   on the GPU box:   tools/ab_binaries.sh run r03 r04 HEAD [rounds]             # -> gpurun_out/ab_rounds.txt
#
# build_ab/ is git-ignored but travels with the gpurun snapshot (like the built .so files).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
mode=${1:-}; shift || true
ARGS="${AB_ARGS:---gpus 1 --steps 5 --warmup 2 --no-cpu --no-legs --dense}"
case $mode in
  prepare)
    for spec in "$@"; do
      tag=${spec%%=*}; rev=${spec#*=}
      rm -rf build_ab/$tag; mkdir -p build_ab/$tag
      git archive "$rev" | tar -x -C build_ab/$tag || exit 1
      rm -rf build_ab/$tag/profiles/history build_ab/$tag/tests build_ab/$tag/gpurun_out
      git rev-parse "$rev" > build_ab/$tag/REV
      make -s -j8 -C build_ab/$tag/owshen_amd/csrc || exit 1
      ls -la build_ab/$tag/owshen_amd/libowshen_gpu.so
    done ;;
  run)
    tags=(); rounds=2
    for a in "$@"; do case $a in [0-9]*) rounds=$a ;; *) tags+=("$a") ;; esac; done
    mkdir -p gpurun_out/ab_rounds
    out=gpurun_out/ab_rounds.txt
    { echo "# same-box A/B of whole rounds, interleaved; args: $ARGS"; echo "# $(date -u +%FT%TZ) $(hostname)"; } > $out
    for r in $(seq 1 $rounds); do
      for tag in "${tags[@]}"; do
        dir=$REPO; [ "$tag" != HEAD ] && dir=$REPO/build_ab/$tag
        ( cd $dir && timeout -s KILL 400 python bench.py $ARGS > $REPO/gpurun_out/ab_rounds/${tag}_$r.json 2> $REPO/gpurun_out/ab_rounds/${tag}_$r.err )
        python - "$tag" "$r" gpurun_out/ab_rounds/${tag}_$r.json <<'PY' | tee -a $out
import json, sys
tag, r, path = sys.argv[1:4]
try:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    iso = d.get("stage_ms_per_step_isolated") or {}
    box = d.get("box") or {}
    print(f"{tag:5s} run {r}: {d['value']:8.2f} proofs/s  {d['ms_per_step']:8.1f} ms/step  isolated acc_g1 {iso.get('accumulate_g1', 0):7.1f} acc_g2 {iso.get('accumulate_g2', 0):7.1f} "
          f"hpoly {iso.get('h_poly', 0):6.1f} reduce {iso.get('reduce_g1', 0) + iso.get('reduce_g2', 0):6.1f} sort {iso.get('digit_sort', 0):6.1f}  sclk {box.get('sclk_MHz')} power {box.get('socket_power_W')}")
except Exception as e:  # noqa: BLE001
    print(f"{tag:5s} run {r}: failed ({e})")
PY
      done
    done ;;
  *) echo "usage: $0 prepare tag=rev ... | run tag ... [rounds]"; exit 2 ;;
esac
