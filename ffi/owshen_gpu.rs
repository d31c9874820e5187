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
#[repr(C)]
pub struct og_r1cs {
    _p: [u8; 0],
}
#[repr(C)]
pub struct og_multi {
    _p: [u8; 0],
}
#[repr(C)]
pub struct og_job {
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
    fn og_pk_windows(pk: *const og_pk, out: *mut u64) -> c_int;
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
    // device memory plumbing + the withdraw circuit: input records -> witnesses -> proofs without leaving HBM
    fn og_malloc(ctx: *mut og_ctx, bytes: usize, out_d: *mut *mut u8) -> c_int;
    fn og_free(ctx: *mut og_ctx, ptr_d: *mut u8) -> c_int;
    fn og_memcpy_h2d(ctx: *mut og_ctx, dst_d: *mut u8, src: *const u8, bytes: usize) -> c_int;
    fn og_memcpy_d2h(ctx: *mut og_ctx, dst: *mut u8, src_d: *const u8, bytes: usize) -> c_int;
    fn og_withdraw_shape(depth: c_int, n_pad3: u64, n_pad2: u64, shape: *mut u64) -> c_int;
    fn og_withdraw_witness_d(ctx: *mut og_ctx, depth: c_int, n_pad3: u64, n_pad2: u64, inputs_d: *const u8, n: usize, witness_out_d: *mut u8) -> c_int;
    fn og_withdraw_prove_batch_d(
        ctx: *mut og_ctx,
        pk: *const og_pk,
        depth: c_int,
        n_pad3: u64,
        n_pad2: u64,
        inputs_d: *const u8,
        n: usize,
        rs: *const u8,
        proofs_out: *mut u8,
        public_out: *mut u8, // n x 6 x 32 B (root, nullifier_hash, recipient, amount, token, chain_id) or null
    ) -> c_int;
    // the same call in two halves: keep one batch ahead of the waits (include/owshen_gpu.h)
    fn og_withdraw_prove_batch_submit_d(
        ctx: *mut og_ctx,
        pk: *const og_pk,
        depth: c_int,
        n_pad3: u64,
        n_pad2: u64,
        inputs_d: *const u8,
        n: usize,
        rs: *const u8,
        proofs_out: *mut u8,
        public_out: *mut u8,
        job_out: *mut *mut og_job,
    ) -> c_int;
    fn og_device_count() -> c_int;
    fn og_job_wait(ctx: *mut og_ctx, job: *mut og_job) -> c_int;
    fn og_job_abandon(ctx: *mut og_ctx, job: *mut og_job) -> c_int;
    fn og_job_poll(ctx: *mut og_ctx, job: *mut og_job, done_out: *mut c_int) -> c_int;
    fn og_mimc7_hash2_d(ctx: *mut og_ctx, left_d: *const u8, right_d: *const u8, out_d: *mut u8, n: usize) -> c_int;
    fn og_mimc7_merkle_paths_d(
        ctx: *mut og_ctx,
        leaves_d: *const u8,
        indices_d: *const u64,
        siblings_d: *const u8,
        depth: c_int,
        nodes_out_d: *mut u8,
        n: usize,
    ) -> c_int;
    // key material
    fn og_withdraw_r1cs(ctx: *mut og_ctx, depth: c_int, n_pad3: u64, n_pad2: u64, dense: c_int, out: *mut *mut og_r1cs) -> c_int;
    fn og_r1cs_free(r1cs: *mut og_r1cs);
    fn og_setup(
        ctx: *mut og_ctx,
        r1cs: *const og_r1cs,
        toxic: *const u8,
        pk_out: *mut *mut u8,
        pk_len: *mut usize,
        vk_out: *mut *mut u8,
        vk_len: *mut usize,
    ) -> c_int;
    fn og_blob_free(blob: *mut u8);
    // the commitment tree fed by mint_tx (src/blockchain/tx/mint_tx.rs:11-49)
    fn og_mimc7_append_d(
        ctx: *mut og_ctx,
        depth: c_int,
        frontier_in_d: *const u8,
        next_index: u64,
        leaves_d: *const u8,
        k: usize,
        frontier_out_d: *mut u8,
        root_out_d: *mut u8,
    ) -> c_int;
    // every GPU of the node from this one process (src/cli/node.rs:71-76 is a single process)
    fn og_multi_init(n_devices: c_int, out: *mut *mut og_multi) -> c_int;
    fn og_multi_shutdown(m: *mut og_multi);
    fn og_multi_size(m: *const og_multi) -> c_int;
    fn og_multi_device_info(m: *const og_multi, rank: c_int, out: *mut u64, pci_out: *mut std::os::raw::c_char) -> c_int;
    fn og_multi_pk_load(m: *mut og_multi, blob: *const u8, len: usize, pks_out: *mut *mut og_pk) -> c_int;
    fn og_multi_pk_free(m: *mut og_multi, pks: *mut *mut og_pk);
    fn og_multi_withdraw_prove_batch(
        m: *mut og_multi,
        pks: *const *mut og_pk,
        depth: c_int,
        n_pad3: u64,
        n_pad2: u64,
        inputs: *const u8,
        n: usize,
        rs: *const u8,
        proofs_out: *mut u8,
        public_out: *mut u8,
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

/// The verifying key as the constructor argument of contracts/WithdrawVerifier.sol: 14 + 2 (n_pub + 1) uint256 words, big-endian,
/// alpha (2) | beta (4) | gamma (4) | delta (4) | IC_0 .. IC_n_pub (2 each), G2 coordinates as (c1, c0) -- the same words as
/// owshen_amd/evm.py `vk_to_evm_words` (tests/test_evm_words.py pins the layout).  `vk`: the "OWVK0001" blob.
pub fn vk_to_evm_words(vk: &[u8]) -> Result<Vec<[u8; 32]>> {
    if vk.len() < 16 || &vk[..8] != b"OWVK0001" {
        return Err(anyhow!("not an OWVK0001 verifying key"));
    }
    let n_pub = u64::from_le_bytes(vk[8..16].try_into().unwrap()) as usize;
    if vk.len() != 16 + 64 + 3 * 128 + (n_pub + 1) * 64 {
        return Err(anyhow!("verifying key length does not match its header"));
    }
    let be = |le: &[u8]| -> [u8; 32] {
        let mut w = [0u8; 32];
        for k in 0..32 {
            w[k] = le[31 - k];
        }
        w
    };
    let mut words = Vec::with_capacity(14 + 2 * (n_pub + 1));
    let g1 = |p: &[u8], out: &mut Vec<[u8; 32]>| {
        out.push(be(&p[0..32]));
        out.push(be(&p[32..64]));
    };
    g1(&vk[16..80], &mut words);
    for k in 0..3 {
        let p = &vk[80 + 128 * k..80 + 128 * (k + 1)]; // x.c0 | x.c1 | y.c0 | y.c1
        for c in [1usize, 0, 3, 2] {
            words.push(be(&p[32 * c..32 * c + 32]));
        }
    }
    for i in 0..=n_pub {
        g1(&vk[464 + 64 * i..464 + 64 * (i + 1)], &mut words);
    }
    Ok(words)
}

/// public inputs as uint256 words (big-endian), the `input` argument of WithdrawVerifier.verifyProof
pub fn public_inputs_to_evm_words(public_inputs: &[Fp]) -> Vec<[u8; 32]> {
    public_inputs
        .iter()
        .map(|x| {
            let le = x.to_repr();
            let mut w = [0u8; 32];
            for k in 0..32 {
                w[k] = le.as_ref()[31 - k];
            }
            w
        })
        .collect()
}

/// GPUs visible to the library (0 without a HIP device): the node's prover task proves on all of them when there are several.
pub fn device_count() -> usize {
    (unsafe { og_device_count() }).max(0) as usize
}

/// Blinding for one proof from the OS CSPRNG.
pub fn random_blinding() -> (Fp, Fp) {
    use ff::Field;
    (Fp::random(rand::rngs::OsRng), Fp::random(rand::rngs::OsRng))
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
// The handles are plain pointers; the library serialises calls on one og_ctx with a mutex of its own (ctx.h: `mu`, taken by
// every entry point), the key is read-only after og_pk_load, and no method here takes `&mut self`.  Sharing one prover as
// `Arc<GpuProver>` across spawn_blocking tasks (which needs `Sync`, not just `Send`) is therefore sound: concurrent calls
// queue inside the library.
unsafe impl Send for GpuProver {}
unsafe impl Sync for GpuProver {}

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
        let mut this = Self { ctx, pk, n_wires: 0, n_pub: 0 }; // from here on Drop releases the key and the context
        let mut info = [0u64; 4];
        check(unsafe { og_pk_info(pk, info.as_mut_ptr()) })?;
        this.n_wires = info[0] as usize;
        this.n_pub = info[1] as usize;
        Ok(this)
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

/// The withdraw statement (oracle/py/withdraw.py): what `withdraw_handler` knows about one request.
/// `index` / `siblings`: the Merkle path of the note's commitment in the depth-`siblings.len()` MiMC7 tree.
#[derive(Clone, Debug)]
pub struct WithdrawRequest {
    pub nullifier: Fp,
    pub secret: Fp,
    pub amount: Fp,
    pub recipient: Fp, // the burn's calldata address as a field element (uint160)
    pub token: Fp,     // the ERC-20 address as a field element, 0 for the native coin (it is part of the note)
    pub chain_id: u64,
    pub index: u64,
    pub siblings: Vec<Fp>,
}

impl WithdrawRequest {
    #[allow(clippy::too_many_arguments)]
    pub fn new(nullifier: Fp, secret: Fp, amount: Fp, recipient: Fp, token: Fp, chain_id: u64, index: u64, siblings: Vec<Fp>) -> Self {
        Self { nullifier, secret, amount, recipient, token, chain_id, index, siblings }
    }

    /// the (8 + depth) x 32-byte input record of og_withdraw_witness_d / og_withdraw_prove_batch_d
    /// (nullifier | secret | amount | recipient | pad_seed | index | token | chain_id | siblings); pad_seed only feeds
    /// synthetic padding gates
    fn record(&self, out: &mut Vec<u8>) {
        for x in [&self.nullifier, &self.secret, &self.amount, &self.recipient] {
            out.extend_from_slice(x.to_repr().as_ref());
        }
        out.extend_from_slice(&[0u8; 32]);
        let mut idx = [0u8; 32];
        idx[..8].copy_from_slice(&self.index.to_le_bytes());
        out.extend_from_slice(&idx);
        out.extend_from_slice(self.token.to_repr().as_ref());
        let mut chain = [0u8; 32];
        chain[..8].copy_from_slice(&self.chain_id.to_le_bytes());
        out.extend_from_slice(&chain);
        for s in &self.siblings {
            out.extend_from_slice(s.to_repr().as_ref());
        }
    }
}

/// Trusted setup from explicit toxic waste (tests / dev nets; production keys come from a ceremony): returns the
/// ("OWPK0001", "OWVK0001") blobs for the depth-`depth` withdraw circuit.
pub fn generate_withdraw_keys(device: i32, depth: i32, toxic: [Fp; 5]) -> Result<(Vec<u8>, Vec<u8>)> {
    unsafe {
        let mut ctx = std::ptr::null_mut();
        check(og_init(device, &mut ctx))?;
        let mut r1cs = std::ptr::null_mut();
        let res = (|| {
            check(og_withdraw_r1cs(ctx, depth, 0, 0, 0, &mut r1cs))?;
            let mut tox = Vec::with_capacity(160);
            for t in &toxic {
                tox.extend_from_slice(t.to_repr().as_ref());
            }
            let (mut pk, mut vk, mut pk_len, mut vk_len) = (std::ptr::null_mut(), std::ptr::null_mut(), 0usize, 0usize);
            check(og_setup(ctx, r1cs, tox.as_ptr(), &mut pk, &mut pk_len, &mut vk, &mut vk_len))?;
            let out = (std::slice::from_raw_parts(pk, pk_len).to_vec(), std::slice::from_raw_parts(vk, vk_len).to_vec());
            og_blob_free(pk);
            og_blob_free(vk);
            Ok(out)
        })();
        if !r1cs.is_null() {
            og_r1cs_free(r1cs);
        }
        og_shutdown(ctx);
        res
    }
}

/// What `prove_withdraw` hands back: the proof and the statement's six public inputs in verifier order
/// (root, nullifier_hash, recipient, amount, token, chain_id) -- root and nullifier_hash are computed on the GPU.
#[derive(Clone, Debug)]
pub struct ProvedWithdraw {
    pub proof: Proof,
    pub root: Fp,
    pub nullifier_hash: Fp,
    pub public: [Fp; 6],
}

/// A submitted batch: owns the host buffers the library fills until `wait` (or drop, which abandons the job: `og_job_abandon`).
pub struct PendingBatch<'a> {
    prover: &'a GpuProver,
    job: *mut og_job,
    recs_d: *mut u8,
    rs: Vec<u8>,
    proofs: Vec<u8>,
    publics: Vec<u8>,
}

impl PendingBatch<'_> {
    /// Has every kernel of the batch finished (`wait` would return at once)?  Non-blocking (`og_job_poll`).
    pub fn is_done(&self) -> Result<bool> {
        let mut done: c_int = 0;
        check(unsafe { og_job_poll(self.prover.ctx, self.job, &mut done) })?;
        Ok(done != 0)
    }

    pub fn wait(mut self) -> Result<Vec<(Proof, [Fp; 6])>> {
        let job = std::mem::replace(&mut self.job, std::ptr::null_mut());
        check(unsafe { og_job_wait(self.prover.ctx, job) })?;
        self.proofs
            .chunks_exact(256)
            .zip(self.publics.chunks_exact(192))
            .map(|(c, p)| {
                let mut public = [Fp::from(0u64); 6];
                for (i, x) in public.iter_mut().enumerate() {
                    *x = fp_from_bytes(&p[32 * i..32 * i + 32])?;
                }
                Ok((Proof(c.try_into().unwrap()), public))
            })
            .collect()
    }
}

impl Drop for PendingBatch<'_> {
    fn drop(&mut self) {
        unsafe {
            if !self.job.is_null() {
                // dropped without `wait`: nothing is copied out, but the job's kernels must finish before our buffers and the
                // records go away, and its call slot must be freed (a slot held by a dropped job would refuse the next submit)
                og_job_abandon(self.prover.ctx, self.job);
            }
            if !self.recs_d.is_null() {
                og_free(self.prover.ctx, self.recs_d);
            }
        }
    }
}

fn fp_from_bytes(b: &[u8]) -> Result<Fp> {
    let mut repr = <Fp as PrimeField>::Repr::default();
    repr.as_mut().copy_from_slice(b);
    Option::<Fp>::from(Fp::from_repr(repr)).ok_or_else(|| anyhow!("library returned a non-canonical field element"))
}

impl GpuProver {
    /// Request -> proof + public inputs: the witness is generated on the GPU (batched MiMC7 path hashing) and never leaves
    /// HBM; the prove call itself returns the public wires (`public_out`), so nothing is computed twice.
    pub fn prove_withdraw(&self, req: &WithdrawRequest, r: Fp, s: Fp) -> Result<ProvedWithdraw> {
        let depth = req.siblings.len() as c_int;
        let mut shape = [0u64; 3];
        check(unsafe { og_withdraw_shape(depth, 0, 0, shape.as_mut_ptr()) })?;
        if shape[0] as usize != self.n_wires || shape[2] != 6 {
            return Err(anyhow!("key is for {} wires, a depth-{} withdraw circuit has {}", self.n_wires, depth, shape[0]));
        }
        let mut rec = Vec::with_capacity((8 + req.siblings.len()) * 32);
        req.record(&mut rec);
        let mut rs = Vec::with_capacity(64);
        rs.extend_from_slice(r.to_repr().as_ref());
        rs.extend_from_slice(s.to_repr().as_ref());
        let mut proof = [0u8; 256];
        let mut publics = [0u8; 192];
        unsafe {
            let mut rec_d = std::ptr::null_mut();
            check(og_malloc(self.ctx, rec.len(), &mut rec_d))?;
            let res = (|| {
                check(og_memcpy_h2d(self.ctx, rec_d, rec.as_ptr(), rec.len()))?;
                check(og_withdraw_prove_batch_d(self.ctx, self.pk, depth, 0, 0, rec_d, 1, rs.as_ptr(), proof.as_mut_ptr(), publics.as_mut_ptr()))
            })();
            og_free(self.ctx, rec_d);
            res?;
        }
        let mut public = [Fp::from(0u64); 6];
        for (i, p) in public.iter_mut().enumerate() {
            *p = fp_from_bytes(&publics[32 * i..32 * i + 32])?;
        }
        Ok(ProvedWithdraw { proof: Proof(proof), root: public[0], nullifier_hash: public[1], public })
    }

    /// A batch in two halves: `submit_withdraw_batch` enqueues everything and returns at once; `PendingBatch::wait` blocks and
    /// yields the proofs with their public inputs.  Submit batch k + 1 before waiting for batch k (at most two in flight).
    pub fn submit_withdraw_batch(&self, reqs: &[WithdrawRequest], rs: &[(Fp, Fp)]) -> Result<PendingBatch<'_>> {
        if reqs.is_empty() || reqs.len() != rs.len() {
            return Err(anyhow!("one (r, s) pair per request"));
        }
        let depth = reqs[0].siblings.len();
        let mut recs = Vec::with_capacity(reqs.len() * (8 + depth) * 32);
        for q in reqs {
            if q.siblings.len() != depth {
                return Err(anyhow!("all requests of a batch must share the tree depth"));
            }
            q.record(&mut recs);
        }
        let mut rsb = Vec::with_capacity(rs.len() * 64);
        for (r, s) in rs {
            rsb.extend_from_slice(r.to_repr().as_ref());
            rsb.extend_from_slice(s.to_repr().as_ref());
        }
        let mut pending = PendingBatch {
            prover: self,
            job: std::ptr::null_mut(),
            recs_d: std::ptr::null_mut(),
            rs: rsb,
            proofs: vec![0u8; reqs.len() * 256],
            publics: vec![0u8; reqs.len() * 192],
        };
        unsafe {
            check(og_malloc(self.ctx, recs.len(), &mut pending.recs_d))?;
            check(og_memcpy_h2d(self.ctx, pending.recs_d, recs.as_ptr(), recs.len()))?;
            check(og_withdraw_prove_batch_submit_d(
                self.ctx,
                self.pk,
                depth as c_int,
                0,
                0,
                pending.recs_d,
                reqs.len(),
                pending.rs.as_ptr(),
                pending.proofs.as_mut_ptr(),
                pending.publics.as_mut_ptr(),
                &mut pending.job,
            ))?;
        }
        Ok(pending)
    }

    /// MultiMiMC7 2-to-1 hash of n pairs on the GPU (og_mimc7_hash2_d): the hash of the commitment tree and of the notes.
    pub fn mimc7_hash2(&self, left: &[Fp], right: &[Fp]) -> Result<Vec<Fp>> {
        if left.len() != right.len() || left.is_empty() {
            return Err(anyhow!("mimc7_hash2: equal, non-zero numbers of left and right inputs"));
        }
        let n = left.len();
        let ser = |v: &[Fp]| v.iter().flat_map(|x| x.to_repr().as_ref().to_vec()).collect::<Vec<u8>>();
        let (l, r) = (ser(left), ser(right));
        let mut out = vec![0u8; n * 32];
        unsafe {
            let mut buf = std::ptr::null_mut();
            check(og_malloc(self.ctx, 3 * n * 32, &mut buf))?;
            let res = (|| {
                check(og_memcpy_h2d(self.ctx, buf, l.as_ptr(), n * 32))?;
                check(og_memcpy_h2d(self.ctx, buf.add(n * 32), r.as_ptr(), n * 32))?;
                check(og_mimc7_hash2_d(self.ctx, buf, buf.add(n * 32), buf.add(2 * n * 32), n))?;
                check(og_memcpy_d2h(self.ctx, out.as_mut_ptr(), buf.add(2 * n * 32), n * 32))
            })();
            og_free(self.ctx, buf);
            res?;
        }
        out.chunks_exact(32).map(fp_from_bytes).collect()
    }

    /// The depth + 1 nodes on the path from `leaf` (at `index`) to the root given its siblings, bottom-up
    /// (og_mimc7_merkle_paths_d): nodes[0] = leaf, nodes[depth] = root.
    pub fn merkle_path_nodes(&self, leaf: Fp, index: u64, siblings: &[Fp]) -> Result<Vec<Fp>> {
        let depth = siblings.len();
        let sib: Vec<u8> = siblings.iter().flat_map(|x| x.to_repr().as_ref().to_vec()).collect();
        let mut out = vec![0u8; (depth + 1) * 32];
        unsafe {
            let mut buf = std::ptr::null_mut();
            let total = 32 + 32 + depth * 32 + (depth + 1) * 32; // leaf | index (8 B, padded) | siblings | nodes
            check(og_malloc(self.ctx, total, &mut buf))?;
            let (leaf_d, idx_d, sib_d, nodes_d) = (buf, buf.add(32), buf.add(64), buf.add(64 + depth * 32));
            let res = (|| {
                check(og_memcpy_h2d(self.ctx, leaf_d, leaf.to_repr().as_ref().as_ptr(), 32))?;
                check(og_memcpy_h2d(self.ctx, idx_d, index.to_le_bytes().as_ptr(), 8))?;
                check(og_memcpy_h2d(self.ctx, sib_d, sib.as_ptr(), depth * 32))?;
                check(og_mimc7_merkle_paths_d(self.ctx, leaf_d, idx_d as *const u64, sib_d, depth as c_int, nodes_d, 1))?;
                check(og_memcpy_d2h(self.ctx, out.as_mut_ptr(), nodes_d, (depth + 1) * 32))
            })();
            og_free(self.ctx, buf);
            res?;
        }
        out.chunks_exact(32).map(fp_from_bytes).collect()
    }

    /// Append `leaves` to the depth-`frontier.len()` commitment tree kept as a frontier (`mint_tx`,
    /// src/blockchain/tx/mint_tx.rs:11-49, would persist the returned frontier + root through KvStore).
    pub fn tree_append(&self, frontier: &[Fp], next_index: u64, leaves: &[Fp]) -> Result<(Vec<Fp>, Fp)> {
        let depth = frontier.len();
        let ser = |v: &[Fp]| v.iter().flat_map(|x| x.to_repr().as_ref().to_vec()).collect::<Vec<u8>>();
        let (fin, lv) = (ser(frontier), ser(leaves));
        let mut out = vec![0u8; depth * 32 + 32];
        unsafe {
            let mut buf = std::ptr::null_mut();
            let total = fin.len() + lv.len() + depth * 32 + 32;
            check(og_malloc(self.ctx, total, &mut buf))?;
            let (fin_d, lv_d, fout_d) = (buf, buf.add(fin.len()), buf.add(fin.len() + lv.len()));
            let res = (|| {
                check(og_memcpy_h2d(self.ctx, fin_d, fin.as_ptr(), fin.len()))?;
                check(og_memcpy_h2d(self.ctx, lv_d, lv.as_ptr(), lv.len()))?;
                check(og_mimc7_append_d(self.ctx, depth as c_int, fin_d, next_index, lv_d, leaves.len(), fout_d, fout_d.add(depth * 32)))?;
                check(og_memcpy_d2h(self.ctx, out.as_mut_ptr(), fout_d, depth * 32 + 32))
            })();
            og_free(self.ctx, buf);
            res?;
        }
        let f = out[..depth * 32].chunks_exact(32).map(fp_from_bytes).collect::<Result<Vec<_>>>()?;
        Ok((f, fp_from_bytes(&out[depth * 32..])?))
    }
}

/// All GPUs of the node behind one handle: proofs are sharded across devices inside the library (replicated key, no
/// data-path collective), which is what a single-process node (src/cli/node.rs:71-76) can call.
pub struct MultiGpuProver {
    m: *mut og_multi,
    pks: Vec<*mut og_pk>,
}
unsafe impl Send for MultiGpuProver {}
unsafe impl Sync for MultiGpuProver {} // og_multi serialises its calls with a mutex of its own (multi.hip)

impl MultiGpuProver {
    pub fn new(n_devices: i32, key: &[u8]) -> Result<Self> {
        let mut m = std::ptr::null_mut();
        check(unsafe { og_multi_init(n_devices, &mut m) })?;
        let n = unsafe { og_multi_size(m) } as usize;
        let mut pks = vec![std::ptr::null_mut(); n];
        if let Err(e) = check(unsafe { og_multi_pk_load(m, key.as_ptr(), key.len(), pks.as_mut_ptr()) }) {
            unsafe { og_multi_shutdown(m) };
            return Err(e);
        }
        Ok(Self { m, pks })
    }

    /// (HIP device ordinal, PCI address, RCCL communicator size) of every rank: what a node logs at start-up
    pub fn devices(&self) -> Result<Vec<(u64, String, u64)>> {
        (0..self.pks.len())
            .map(|r| {
                let mut out = [0u64; 4];
                let mut pci = [0 as std::os::raw::c_char; 32];
                check(unsafe { og_multi_device_info(self.m, r as c_int, out.as_mut_ptr(), pci.as_mut_ptr()) })?;
                let s = unsafe { CStr::from_ptr(pci.as_ptr()) }.to_string_lossy().into_owned();
                Ok((out[0], s, out[1]))
            })
            .collect()
    }

    /// proofs + the six public inputs of every proof (root, nullifier_hash, recipient, amount, token, chain_id)
    pub fn prove_withdraw_batch(&self, reqs: &[WithdrawRequest], rs: &[(Fp, Fp)]) -> Result<Vec<(Proof, [Fp; 6])>> {
        if reqs.is_empty() || reqs.len() != rs.len() {
            return Err(anyhow!("one (r, s) pair per request"));
        }
        let depth = reqs[0].siblings.len();
        let mut recs = Vec::with_capacity(reqs.len() * (8 + depth) * 32);
        for q in reqs {
            if q.siblings.len() != depth {
                return Err(anyhow!("all requests of a batch must share the tree depth"));
            }
            q.record(&mut recs);
        }
        let mut rsb = Vec::with_capacity(rs.len() * 64);
        for (r, s) in rs {
            rsb.extend_from_slice(r.to_repr().as_ref());
            rsb.extend_from_slice(s.to_repr().as_ref());
        }
        let mut out = vec![0u8; reqs.len() * 256];
        let mut pubs = vec![0u8; reqs.len() * 192];
        check(unsafe {
            og_multi_withdraw_prove_batch(
                self.m,
                self.pks.as_ptr(),
                depth as c_int,
                0,
                0,
                recs.as_ptr(),
                reqs.len(),
                rsb.as_ptr(),
                out.as_mut_ptr(),
                pubs.as_mut_ptr(),
            )
        })?;
        out.chunks_exact(256)
            .zip(pubs.chunks_exact(192))
            .map(|(c, p)| {
                let mut public = [Fp::from(0u64); 6];
                for (i, x) in public.iter_mut().enumerate() {
                    *x = fp_from_bytes(&p[32 * i..32 * i + 32])?;
                }
                Ok((Proof(c.try_into().unwrap()), public))
            })
            .collect()
    }
}

impl Drop for MultiGpuProver {
    fn drop(&mut self) {
        unsafe {
            og_multi_pk_free(self.m, self.pks.as_mut_ptr());
            og_multi_shutdown(self.m);
        }
    }
}
