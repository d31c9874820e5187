#!/bin/bash
# TEST INFRASTRUCTURE: builds the CPU-interpreted kernels (tests/hipemu) with UBSan and ASan and runs the shared parity
# cases on them.  Every "device" buffer is a malloc block there, so an out-of-bounds index in a kernel or in the host
# glue is a hard ASan error; signed-overflow / shift bugs in the limb arithmetic are UBSan errors.
#   tools/sanitize_emu.sh        (about 15 minutes: most of it compiling the kernels with the sanitizers)
#   tools/sanitize_emu.sh full   (after the above: the WHOLE interpreter suite, tests/test_emu_*.py, on both sanitized builds -- tests/emu.py
#                                 loads $OG_EMU_LIB instead of building its own; about 20 minutes per sanitizer)
set -e
cd "$(dirname "$0")/../tests/hipemu"
CS=../../owshen_amd/csrc
for san in undefined address; do
  out=/tmp/og_san_$san; mkdir -p $out
  flags="-O1 -g -std=c++17 -fPIC -DOG_AB_HOOKS -fsanitize=$san -I. -I$CS -Wno-attributes -Wno-unknown-pragmas"  # (the hooks build, like tests/hipemu/Makefile: the cases reach rare paths at toy sizes through OG_* switches)
  [ $san = undefined ] && flags="$flags -fno-sanitize-recover=undefined"
  ( for f in capi field_ops mimc7 msm msm_g1 msm_g2 ntt groth16 witness verify multi keygen eddsa zkey; do echo "g++ $flags -x c++ -c $CS/$f.hip -o $out/$f.o"; done
    echo "g++ $flags -c $CS/keccak_host.cpp -o $out/keccak.o"; echo "g++ $flags -c emu_runtime.cpp -o $out/emu_runtime.o"
    echo "g++ $flags -c stubs.cpp -o $out/stubs.o" ) | xargs -P 8 -I{} sh -c "{}"
  g++ -shared -fPIC -fsanitize=$san $out/*.o -o $out/libowshen_emu_san.so
  lib=$(gcc -print-file-name=lib$([ $san = undefined ] && echo ubsan || echo asan).so)
  echo "== $san"
  ( cd ../.. && OG_EMU_LIB=$out/libowshen_emu_san.so UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
      ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1 LD_PRELOAD=$lib python - <<'PY'
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import tests.emu as emu
from owshen_amd._abi import bind
emu.lib = bind(C.CDLL(os.environ["OG_EMU_LIB"]))
emu.Ctx._lib = emu.lib
from tests import groth16_cases as gc, withdraw_cases as wc, tree_cases as tc, golden_cases as goldc
c = emu.Ctx()
goldc.check_device(c)
gc.case_prove_batch_matches_oracle_and_verifies(c)
os.environ["OG_G2_AFFINE"] = "1"   # the G2 buckets by batched affine additions (k_accumulate_affine)
gc.case_prove_batch_matches_oracle_and_verifies(c)
del os.environ["OG_G2_AFFINE"]
gc.case_noncanonical_witness_is_rejected(c)
gc.case_degenerate_circuits(c)
gc.case_random_shapes(c, range(3000, 3004))
wc.case_r1cs_and_witness_match_spec(c, 3, 7, 130)
wc.case_host_chains_gives_the_kernels_bytes(c, 2, 2, 3)   # og_set_host_chains: the host's walk (pinned staging, a thread per request) and assembly
# the stage pipeline (ramped plan, two scratch slots released in two steps, persistent launches) and calls kept one ahead, at toy size
os.environ.update(OG_SUB_BATCH="2", OG_PIPE_MIN="1", OG_GEN_MIN="1")
gc.case_medium_circuit_vs_c_oracle(c, 60, 9, None)
wc.case_submitted_batches_equal_blocking_calls(c, 1, 2, 3, [5, 2, 4], third_is_refused=True)
for k in ("OG_SUB_BATCH", "OG_PIPE_MIN", "OG_GEN_MIN"):
    del os.environ[k]
tc.case_append_matches_incremental_tree(c, 5, [7, 1, 8], 1)
class Env:  # what the case needs of pytest's monkeypatch
    def setenv(self, k, v): os.environ[k] = v
tc.case_one_and_two_lanes_per_hash_agree(c, Env(), n_hash=7, n_paths=3, depth=4, n_leaves=16, witness_depth=2)
del os.environ["OG_MIMC_PAIR"]
os.environ.pop("OG_WITNESS_W9", None); os.environ.pop("OG_MIMC_W9", None); os.environ.pop("OG_W9_ROWS", None)
c.close()
print("clean")
PY
  )
done
if [ "$1" = zkey ]; then   # only the file parsers and the DFT over points (the part of the library that reads bytes from outside)
  cd ../..
  for san in undefined address; do
    lib=$(gcc -print-file-name=lib$([ $san = undefined ] && echo ubsan || echo asan).so)
    echo "== snarkjs files, $san"
    OG_EMU_LIB=/tmp/og_san_$san/libowshen_emu_san.so UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
      ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1 LD_PRELOAD=$lib \
      python -m pytest tests/test_emu_zkey.py tests/test_zkey_fuzz.py -x -q -p no:cacheprovider
  done
fi
if [ "$1" = full ]; then
  cd ../..
  for san in undefined address; do
    lib=$(gcc -print-file-name=lib$([ $san = undefined ] && echo ubsan || echo asan).so)
    echo "== full suite, $san"
    OG_EMU_LIB=/tmp/og_san_$san/libowshen_emu_san.so UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
      ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1 LD_PRELOAD=$lib \
      python -m pytest tests/test_emu_field29.py tests/test_emu_w9.py tests/test_emu_kernels.py tests/test_emu_groth16.py tests/test_emu_withdraw.py tests/test_emu_tree.py \
        tests/test_emu_eddsa.py tests/test_emu_multi.py tests/test_emu_multi8.py tests/test_emu_deposit.py tests/test_emu_zkey.py tests/test_zkey_fuzz.py -x -q -p no:cacheprovider
  done
fi
