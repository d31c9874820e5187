//! Rust binding for libowshen_gpu.so (include/owshen_gpu.h) -- SOURCE ONLY.
//!
//! This image has no `rustc`/`cargo`, so this file is not compiled here; it is the shim a maintainer drops
//! into the reference crate as `src/prover/owshen_gpu.rs`.  It keeps the crate's
//! conventions: `anyhow::Result` errors (src/utils.rs:5-20), `Fp` field elements serialised with
//! `to_repr()` = 32-byte little-endian canonical (src/blockchain/tx/owshen_airdrop/babyjubjub/mod.rs:7-11).
//! The reference has no prove()/verify() today; these names are this build's proposal.
#![allow(non_camel_case_types)]
use anyhow::{anyhow, Result};
use ff::PrimeField;
use std::ffi::CStr;
use std::os::raw::{c_char, c_int};

use crate::blockchain::tx::owshen_airdrop::babyjubjub::Fp;

#[repr(C)]
pub struct og_ctx {
    _p: [u8; 0],
}
#[repr(C)]
pub struct og_pk {
    _p: [u8; 0],
}

#[link(name = "owshen_gpu")]
extern "C" {
    fn og_init(device: c_int, out: *mut *mut og_ctx) -> c_int;
    fn og_shutdown(ctx: *mut og_ctx);
    fn og_last_error() -> *const c_char;
    fn og_pk_load(ctx: *mut og_ctx, blob: *const u8, len: usize, out: *mut *mut og_pk) -> c_int;
    fn og_pk_free(pk: *mut og_pk);
    fn og_pk_info(pk: *const og_pk, info: *mut u64) -> c_int;
    fn og_prove(ctx: *mut og_ctx, pk: *const og_pk, witness: *const u8, rs: *const u8, proof_out: *mut u8) -> c_int;
    fn og_verify(vk: *const u8, vk_len: usize, public_inputs: *const u8, n_pub: usize, proof: *const u8, ok_out: *mut c_int) -> c_int;
    fn og_prove_batch(
        ctx: *mut og_ctx,
        pk: *const og_pk,
        witnesses: *const u8,
        n: usize,
        rs: *const u8,
        proofs_out: *mut u8,
    ) -> c_int;
}

fn check(rc: c_int) -> Result<()> {
    if rc == 0 {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(og_last_error()) }.to_string_lossy().into_owned();
    Err(anyhow!("owshen_gpu error {}: {}", rc, msg))
}

/// A = G1 (64 B) || B = G2 (128 B) || C = G1 (64 B), affine, 32-byte little-endian coordinates.
#[derive(Clone, Debug, PartialEq, Eq)]
pub struct Proof(pub [u8; 256]);

impl Proof {
    /// 8 x uint256 big-endian, G2 as (x.c1, x.c0, y.c1, y.c0): the calldata of a snarkjs-style verifier.
    pub fn to_evm_calldata(&self) -> [u8; 256] {
        let mut out = [0u8; 256];
        let order = [0usize, 1, 3, 2, 5, 4, 6, 7];
        for (dst, src) in order.iter().enumerate() {
            for k in 0..32 {
                out[dst * 32 + k] = self.0[src * 32 + 31 - k];
            }
        }
        out
    }
}

/// Groth16 verification on the CPU (no GPU, no context): `vk` is the "OWVK0001" blob.  `Ok(false)` = the proof does
/// not verify (including malformed proof encodings); `Err` = the verifying key itself is malformed.
pub fn verify(vk: &[u8], public_inputs: &[Fp], proof: &Proof) -> Result<bool> {
    let mut pubs = Vec::with_capacity(public_inputs.len() * 32);
    for x in public_inputs {
        pubs.extend_from_slice(x.to_repr().as_ref());
    }
    let mut ok: c_int = 0;
    check(unsafe { og_verify(vk.as_ptr(), vk.len(), pubs.as_ptr(), public_inputs.len(), proof.0.as_ptr(), &mut ok) })?;
    Ok(ok == 1)
}

/// One GPU context (one per process / per GPU).  Calls are blocking: wrap them in
/// `tokio::task::spawn_blocking` and do NOT hold the `Context` mutex across them.
pub struct GpuProver {
    ctx: *mut og_ctx,
    pk: *mut og_pk,
    pub n_wires: usize,
    pub n_pub: usize,
}
unsafe impl Send for GpuProver {}

impl GpuProver {
    /// `key`: the serialized proving key ("OWPK0001", see include/owshen_gpu.h); parsed once, then resident in HBM.
    pub fn new(device: i32, key: &[u8]) -> Result<Self> {
        let mut ctx = std::ptr::null_mut();
        check(unsafe { og_init(device, &mut ctx) })?;
        let mut pk = std::ptr::null_mut();
        if let Err(e) = check(unsafe { og_pk_load(ctx, key.as_ptr(), key.len(), &mut pk) }) {
            unsafe { og_shutdown(ctx) };
            return Err(e);
        }
        let mut info = [0u64; 4];
        check(unsafe { og_pk_info(pk, info.as_mut_ptr()) })?;
        Ok(Self { ctx, pk, n_wires: info[0] as usize, n_pub: info[1] as usize })
    }

    /// witness[0] must be Fp::ONE, witness[1..=n_pub] the public inputs.  (r, s): the caller's blinding,
    /// explicit so that a proof is reproducible; draw them from a CSPRNG in production.
    pub fn prove(&self, witness: &[Fp], r: Fp, s: Fp) -> Result<Proof> {
        Ok(self.prove_batch(&[witness], &[(r, s)])?.remove(0))
    }

    pub fn prove_batch(&self, witnesses: &[&[Fp]], rs: &[(Fp, Fp)]) -> Result<Vec<Proof>> {
        if witnesses.len() != rs.len() {
            return Err(anyhow!("one (r, s) pair per witness"));
        }
        let mut w = Vec::with_capacity(witnesses.len() * self.n_wires * 32);
        for z in witnesses {
            if z.len() != self.n_wires {
                return Err(anyhow!("witness has {} wires, key expects {}", z.len(), self.n_wires));
            }
            for x in z.iter() {
                w.extend_from_slice(x.to_repr().as_ref());
            }
        }
        let mut rsb = Vec::with_capacity(rs.len() * 64);
        for (r, s) in rs {
            rsb.extend_from_slice(r.to_repr().as_ref());
            rsb.extend_from_slice(s.to_repr().as_ref());
        }
        let mut out = vec![0u8; witnesses.len() * 256];
        check(unsafe { og_prove_batch(self.ctx, self.pk, w.as_ptr(), witnesses.len(), rsb.as_ptr(), out.as_mut_ptr()) })?;
        Ok(out
            .chunks_exact(256)
            .map(|c| {
                let mut p = [0u8; 256];
                p.copy_from_slice(c);
                Proof(p)
            })
            .collect())
    }
}

impl Drop for GpuProver {
    fn drop(&mut self) {
        unsafe {
            og_pk_free(self.pk);
            og_shutdown(self.ctx);
        }
    }
}
