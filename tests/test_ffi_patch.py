"""The Rust side of the seam cannot be compiled here (no rustc), so what CAN be checked is checked:
ffi/burn_proof.patch applies to the reference checkout with `patch -p1` (round 2's did not), is the output of its generator,
everything the patched files call in `crate::prover` is defined by the patch or by ffi/owshen_gpu.rs, and the shim's
extern "C" block declares every function with the argument count of include/owshen_gpu.h."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
PATCH = os.path.join(ROOT, "ffi", "burn_proof.patch")
SHIM = os.path.join(ROOT, "ffi", "owshen_gpu.rs")

needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="the reference checkout exists only in the build container")


def _patched_tree(tmp_path):
    text = open(PATCH).read()
    touched = sorted(set(re.findall(r"^\+\+\+ b/(\S+)", text, flags=re.M)))
    for rel in touched:
        src = os.path.join(REF, rel)
        if os.path.exists(src):                      # new files (src/prover/mod.rs, build.rs) have no original
            dst = tmp_path / rel
            dst.parent.mkdir(parents=True, exist_ok=True)
            shutil.copyfile(src, dst)
    return touched


@needs_ref
def test_patch_applies_with_patch_p1(tmp_path):
    touched = _patched_tree(tmp_path)
    dry = subprocess.run(["patch", "-p1", "--dry-run", "-i", PATCH], cwd=tmp_path, capture_output=True, text=True)
    assert dry.returncode == 0, dry.stdout + dry.stderr
    real = subprocess.run(["patch", "-p1", "-i", PATCH], cwd=tmp_path, capture_output=True, text=True)
    assert real.returncode == 0, real.stdout + real.stderr
    assert "src/prover/mod.rs" in touched and (tmp_path / "src/prover/mod.rs").exists() and (tmp_path / "build.rs").exists()
    assert not list(tmp_path.rglob("*.rej")) and not list(tmp_path.rglob("*.orig"))
    # every Burn / Mint / ERC20 literal of the touched files names exactly the fields of the (patched) struct -- at the
    # literal's own brace depth: `commitment` inside a nested `ERC20 { .. }` would not compile
    def struct_fields(text, name):
        body = re.search(r"pub struct %s \{(.*?)\n\}" % name, text, flags=re.S).group(1)
        return set(re.findall(r"^\s*pub (\w+):", body, flags=re.M))

    def top_level_fields(body):
        parts, depth, cur = [], 0, ""
        for ch in re.sub(r"//[^\n]*", "", body):      # split at the commas of the literal's own depth
            depth += {"{": 1, "(": 1, "[": 1, "}": -1, ")": -1, "]": -1}.get(ch, 0)
            if ch == "," and depth == 0:
                parts.append(cur)
                cur = ""
            else:
                cur += ch
        parts.append(cur)
        return {re.match(r"\s*(\w+)", part).group(1) for part in parts if part.strip()}   # `name: value` or shorthand `name`

    custom = (tmp_path / "src/types/tx/custom.rs").read_text()
    fields = {"Burn": struct_fields(custom, "Burn"), "Mint": struct_fields(custom, "Mint"),
              "ERC20": struct_fields(open(os.path.join(REF, "src/types/mod.rs")).read(), "ERC20")}
    assert "proof" in fields["Burn"] and "commitment" in fields["Mint"] and "commitment" not in fields["ERC20"]
    n_lit = {"Burn": 0, "Mint": 0, "ERC20": 0}
    for rel in touched:
        if not rel.endswith(".rs") or rel.startswith("src/prover"):
            continue
        s = (tmp_path / rel).read_text()
        for m in re.finditer(r"\b(Burn|Mint|ERC20) \{\s*\n", s):
            if re.search(r"pub struct|impl |->", s[max(0, s.rfind("\n", 0, m.start())):m.end()]):
                continue
            depth, k = 1, m.end()
            while depth:                                  # the literal's closing brace (fields may nest braces)
                depth += {"{": 1, "}": -1}.get(s[k], 0)
                k += 1
            body = s[m.end():k - 1]
            if ".." in re.sub(r"//[^\n]*", "", body) and "..=" not in body:
                continue                                  # struct-update syntax names the rest
            got = top_level_fields(body)
            assert got == fields[m.group(1)], f"{rel}: a {m.group(1)} literal with fields {sorted(got)}, the struct has {sorted(fields[m.group(1)])}"
            n_lit[m.group(1)] += 1
    assert n_lit["Burn"] >= 8 and n_lit["Mint"] >= 14, n_lit
    # new Key variants sit after the last upstream variant (bincode encodes the variant index)
    key = (tmp_path / "src/db/key.rs").read_text()
    assert key.index("TokenSymbol(Address)") < key.index("NoteTreeFrontier") < key.index("WithdrawVk")


@needs_ref
def test_patch_is_the_generators_output():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_ffi_patch.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def _defined_names(rust):
    return set(re.findall(r"\b(?:pub\s+)?(?:fn|struct|const|static|mod)\s+([A-Za-z_][A-Za-z0-9_]*)", rust))


def test_everything_the_patch_calls_is_defined():
    """identifiers used as prover::X / notes::X / gpu.method(..) in the added lines exist in src/prover/mod.rs (part of the
    patch) or in the shim"""
    text = open(PATCH).read()
    added = "\n".join(ln[1:] for ln in text.splitlines() if ln.startswith("+") and not ln.startswith("+++"))
    mod_rs = "\n".join(ln[1:] for ln in text[text.index("+++ b/src/prover/mod.rs"):].split("\n--- ")[0].splitlines() if ln.startswith("+"))
    shim = open(SHIM).read()
    defined = _defined_names(mod_rs) | _defined_names(shim)
    used = set(re.findall(r"\bprover::([A-Za-z_][A-Za-z0-9_]*)", added)) | set(re.findall(r"\bnotes::([A-Za-z_][A-Za-z0-9_]*)", added))
    used -= {"self", "notes"}
    missing = sorted(u for u in used if u not in defined)
    assert not missing, f"called but never defined: {missing}"
    # methods called on the prover handle
    for meth in set(re.findall(r"\b(?:gpu|p)\.([a-z_0-9]+)\(", added)):
        assert re.search(r"\bfn %s\b" % meth, shim), f"GpuProver::{meth} is not in ffi/owshen_gpu.rs"
    # fields read from the prove result
    for field in set(re.findall(r"\bproved\.([a-z_]+)", added)):
        assert re.search(r"pub %s\s*:" % field, shim), f"ProvedWithdraw.{field}"
    assert "unsafe impl Sync for GpuProver" in shim          # Arc<GpuProver> crosses spawn_blocking
    assert "pub fn vk_to_evm_words" in shim and "pub fn new(" in shim


def _c_arg_count(header, name):
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)       # declarations only, not the prose around them
    m = re.search(r"^[A-Za-z_][A-Za-z0-9_ \*]*?\b%s\s*\(([^;{]*?)\)\s*;" % re.escape(name), header, flags=re.S | re.M)
    assert m, f"{name} is not declared in include/owshen_gpu.h"
    args = m.group(1).strip()
    return 0 if args in ("", "void") else args.count(",") + 1


def test_shim_extern_block_matches_the_header():
    header = open(os.path.join(ROOT, "include", "owshen_gpu.h")).read()
    shim = open(SHIM).read()
    block = shim[shim.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    decls = re.findall(r"\bfn (og_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*[^;]+)?;", block, flags=re.S)
    assert len(decls) >= 25
    for name, args in decls:
        args = re.sub(r"//[^\n]*", "", args).strip().rstrip(",")
        n = 0 if not args else args.count(",") + 1
        assert n == _c_arg_count(header, name), f"{name}: the shim declares {n} arguments, the header {_c_arg_count(header, name)}"


@needs_ref
def test_the_node_runs_the_request_coalescer(tmp_path):
    """the batch-1024 headline has a call site.  `withdraw_handler` hands its request to
    `prover::coalescer::prove`, `run_node` joins `prover::coalescer::run` in its try_join!, and the task proves BATCHES through
    the shim's submit / wait pair (one GPU, one call kept ahead) or MultiGpuProver (several GPUs) -- every name defined."""
    _patched_tree(tmp_path)
    assert subprocess.run(["patch", "-p1", "-i", PATCH], cwd=tmp_path, capture_output=True).returncode == 0
    node = (tmp_path / "src/cli/node.rs").read_text()
    handler = (tmp_path / "src/services/api_services/withdraw.rs").read_text()
    mod_rs = (tmp_path / "src/prover/mod.rs").read_text()
    shim = open(SHIM).read()
    assert "crate::prover::coalescer::run(" in node
    assert re.search(r"tokio::try_join!\(block_producer_fut, api_server_fut, rpc_server_fut, prover_fut\)", node)
    assert "crate::prover::install_multi(&key)" in node
    assert "prover::coalescer::prove(request).await?" in handler and "spawn_blocking" not in handler
    # the handler no longer holds the Context mutex while the proof is made: the lock is taken again AFTER the await
    assert handler.index("coalescer::prove(request).await") < handler.index("let mut _ctx = ctx.lock().await;")
    coal = mod_rs[mod_rs.index("pub mod coalescer {"):mod_rs.index("pub mod notes {")]
    for name in ("pub async fn prove(", "pub async fn run<", "MAX_BATCH: usize = 1024", "spawn_blocking", "submit_withdraw_batch(", ".wait()",
                 "prove_withdraw_batch(", "timeout_at(deadline", "try_recv()", "in_flight"):
        assert name in coal, name
    for fn in ("submit_withdraw_batch", "prove_withdraw_batch", "prove_withdraw", "device_count", "is_done", "wait"):
        assert re.search(r"pub fn %s\b" % fn, shim), fn
    for fn in ("install_multi", "global_multi", "global", "install"):
        assert re.search(r"pub fn %s\b" % fn, mod_rs), fn
    assert "fn og_job_poll(" in shim and "fn og_device_count(" in shim
    assert "derive(Clone" in shim[shim.index("pub struct WithdrawRequest") - 40:shim.index("pub struct WithdrawRequest")]
    # the drain rule the C replay measures is the one the task runs (tools/coalescer.c, tools/coalescer.py)
    csrc = open(os.path.join(ROOT, "tools", "coalescer.c")).read()
    assert "og_withdraw_prove_batch_submit_d" in csrc and "og_job_poll" in csrc and "window_ns" in csrc and "max_batch" in csrc
