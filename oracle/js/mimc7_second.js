// TEST INFRASTRUCTURE -- a SECOND, independent restatement of circomlib's MiMC7 (src/mimc7.js, circuits/mimc.circom), written
// from the circomlib text alone, in another language and on another big-integer engine (V8 BigInt instead of CPython ints),
// with its own Keccak-f[1600].  It shares no code with oracle/py or oracle/c: tests/test_second_engine.py runs it under
// `node` and compares what it prints with the Python oracle, so the constants, the permutation, MultiMiMC7 and the depth-32
// zero-hash chain are no longer "one author, one big-int engine".  Never imported by the product.
//
// circomlib text followed here:
//   SEED = "mimc", NROUNDS = 91
//   getConstants: c = keccak256(SEED); for i = 1 .. 90: c = keccak256(c) (the RAW 32-byte hash is re-hashed),
//                 cts[i] = c mod p; cts[0] = 0
//   hash(x, k):   r = (x + k)^7 for round 0, r = (r + k + cts[i])^7 after; result r + k
//   multiHash(arr, key): r = key; for each a: r = r + a + hash(a, r)
'use strict';
const P = 21888242871839275222246405745257275088548364400416034343698204186575808495617n;

// ---- Keccak-256 (original Keccak padding 0x01, not SHA-3's 0x06), 64-bit lanes as BigInt -------------------------------
const M64 = (1n << 64n) - 1n;
const RC = [];
(function () { // round constants from the LFSR of the Keccak reference
  let r = 1;
  for (let round = 0; round < 24; round++) {
    let c = 0n;
    for (let j = 0; j < 7; j++) {
      if (r & 1) c ^= 1n << BigInt((1 << j) - 1);
      r = (r << 1) ^ ((r >> 7) ? 0x171 : 0);
      r &= 0x1ff; if (r & 0x100) r ^= 0x100;
    }
    RC.push(c);
  }
})();
// the loop above is the textbook rc[t] LFSR x^8 + x^6 + x^5 + x^4 + 1; to be safe it is checked against the first and last
// published constants below (a wrong LFSR would make every hash wrong, which the known answers at the bottom also catch)
const ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]; // [x][y]
function rol(v, n) { n = BigInt(n); return n === 0n ? v : ((v << n) | (v >> (64n - n))) & M64; }
function keccakF(A) { // A[x][y]
  for (let rnd = 0; rnd < 24; rnd++) {
    const C = [], D = [];
    for (let x = 0; x < 5; x++) C.push(A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4]);
    for (let x = 0; x < 5; x++) D.push(C[(x + 4) % 5] ^ rol(C[(x + 1) % 5], 1));
    for (let x = 0; x < 5; x++) for (let y = 0; y < 5; y++) A[x][y] ^= D[x];
    const B = [[], [], [], [], []];
    for (let x = 0; x < 5; x++) for (let y = 0; y < 5; y++) B[y][(2 * x + 3 * y) % 5] = rol(A[x][y], ROT[x][y]);
    for (let x = 0; x < 5; x++) for (let y = 0; y < 5; y++) A[x][y] = B[x][y] ^ ((~B[(x + 1) % 5][y] & M64) & B[(x + 2) % 5][y]);
    A[0][0] ^= RC[rnd];
  }
}
function keccak256(bytes) { // Uint8Array -> Uint8Array(32)
  const rate = 136;
  const padded = new Uint8Array(Math.ceil((bytes.length + 1) / rate) * rate);
  padded.set(bytes);
  padded[bytes.length] ^= 0x01;
  padded[padded.length - 1] ^= 0x80;
  const A = [];
  for (let x = 0; x < 5; x++) A.push([0n, 0n, 0n, 0n, 0n]);
  for (let off = 0; off < padded.length; off += rate) {
    for (let i = 0; i < rate / 8; i++) {
      let lane = 0n;
      for (let b = 7; b >= 0; b--) lane = (lane << 8n) | BigInt(padded[off + 8 * i + b]);
      A[i % 5][Math.floor(i / 5)] ^= lane;
    }
    keccakF(A);
  }
  const out = new Uint8Array(32);
  for (let i = 0; i < 4; i++) {
    let lane = A[i % 5][Math.floor(i / 5)];
    for (let b = 0; b < 8; b++) { out[8 * i + b] = Number(lane & 0xffn); lane >>= 8n; }
  }
  return out;
}
function hex(u8) { return Array.from(u8).map(b => b.toString(16).padStart(2, '0')).join(''); }
function beInt(u8) { let v = 0n; for (const b of u8) v = (v << 8n) | BigInt(b); return v; }

// ---- MiMC7 ---------------------------------------------------------------------------------------------------------------
function constants() {
  const cts = [0n];
  let c = keccak256(Buffer.from('mimc', 'ascii'));
  for (let i = 1; i < 91; i++) {
    c = keccak256(c);
    cts.push(beInt(c) % P);
  }
  return cts;
}
const CTS = constants();
function pow7(t) { const t2 = t * t % P, t4 = t2 * t2 % P; return t4 * t2 % P * t % P; }
function hash(x, k) {
  let r = 0n;
  for (let i = 0; i < 91; i++) {
    const t = i === 0 ? (x + k) % P : (r + k + CTS[i]) % P;
    r = pow7(t);
  }
  return (r + k) % P;
}
function multiHash(arr, key) {
  let r = key === undefined ? 0n : key;
  for (const a of arr) r = (r + a + hash(a, r)) % P;
  return r;
}

const out = {
  engine: 'node ' + process.version + ' BigInt',
  keccak_empty: hex(keccak256(new Uint8Array(0))),
  keccak_abc: hex(keccak256(Buffer.from('abc', 'ascii'))),
  keccak_mimc: hex(keccak256(Buffer.from('mimc', 'ascii'))),
  rc0: RC[0].toString(16), rc23: RC[23].toString(16),
  c1: CTS[1].toString(), c2: CTS[2].toString(), c90: CTS[90].toString(),
  constants_sum: (CTS.reduce((a, b) => a + b, 0n) % P).toString(),
  hash_1_2: hash(1n, 2n).toString(),
  multihash_1_2: multiHash([1n, 2n], 0n).toString(),
  multihash_12_45_78_41: multiHash([12n, 45n, 78n, 41n], 0n).toString(),
  zeros: (function () { const z = [0n]; for (let h = 0; h < 32; h++) z.push(multiHash([z[h], z[h]], 0n)); return z.map(String); })(),
};
console.log(JSON.stringify(out));
