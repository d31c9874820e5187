// TEST INFRASTRUCTURE -- a SECOND, independent BN254 pairing and Groth16 verifier, on another big-integer engine (V8 BigInt)
// and BY ANOTHER ROUTE than oracle/py/pairing.py, so that a whole proof of the product is checked by something that shares
// neither code nor representation with the oracles:
//   * Fp12 is ONE flat extension Fp[w] / (w^12 - 18 w^6 + 82) with schoolbook polynomial arithmetic (the Python oracle uses
//     the tower Fp2 -> Fp6 -> Fp12 with Karatsuba-style formulas); inverses come from Gaussian elimination on the
//     multiplication matrix;
//   * the G2 point is UNTWISTED into E(Fp12) (x -> x' w^2, y -> y' w^3 with u = w^6 - 9) and the Miller loop runs the
//     textbook chord-and-tangent rule there, lines evaluated at the G1 point cast into Fp12 (the Python oracle stays on the
//     twist and assembles sparse line values by hand);
//   * the Frobenius steps of the optimal ate pairing are honest p-th powers in Fp12 (the oracle multiplies by precomputed
//     twist constants);
//   * the final exponentiation is plain square-and-multiply by (p^12 - 1) / r.
// Followed text: the optimal ate pairing on BN curves (Vercauteren), loop 6x + 2 = 29793968203157093288 followed by the
// lines through pi(Q) and -pi^2(Q); EIP-197 for the curve constants and the verification equation; Groth16 section 3.2.
// tests/test_second_engine.py feeds it a verifying key, public inputs and proofs as JSON on stdin and compares e(G1, G2)
// coefficient by coefficient with the Python oracle (after mapping its tower basis onto powers of w).  Never imported by the
// product.
'use strict';
const P = 21888242871839275222246405745257275088696311157297823662689037894645226208583n;
const R = 21888242871839275222246405745257275088548364400416034343698204186575808495617n;
const ATE = 29793968203157093288n;

const fp = (a) => { a %= P; return a < 0n ? a + P : a; };
function fpInv(a) {  // extended Euclid
  let [r0, r1, s0, s1] = [P, fp(a), 0n, 1n];
  if (r1 === 0n) throw new Error('inverse of zero');
  while (r1 !== 0n) { const q = r0 / r1; [r0, r1] = [r1, r0 - q * r1]; [s0, s1] = [s1, s0 - q * s1]; }
  return fp(s0);
}

// ---- G1 over Fp (affine; null = infinity): only the public-input combination needs it ---------------------------------
function g1Add(a, b) {
  if (a === null) return b;
  if (b === null) return a;
  let m;
  if (a[0] === b[0]) {
    if (fp(a[1] + b[1]) === 0n) return null;
    m = fp(3n * a[0] * a[0] * fpInv(2n * a[1]));
  } else m = fp((b[1] - a[1]) * fpInv(b[0] - a[0]));
  const x = fp(m * m - a[0] - b[0]);
  return [x, fp(m * (a[0] - x) - a[1])];
}
function g1Mul(p, k) {
  let acc = null;
  for (let i = k.toString(2).length - 1; i >= 0; i--) { acc = g1Add(acc, acc); if ((k >> BigInt(i)) & 1n) acc = g1Add(acc, p); }
  return acc;
}
const g1OnCurve = (p) => fp(p[1] * p[1] - p[0] * p[0] * p[0] - 3n) === 0n;

// ---- Fp12 = Fp[w] / (w^12 - 18 w^6 + 82): arrays of 12 coefficients ----------------------------------------------------
const ZERO12 = () => new Array(12).fill(0n);
const ONE12 = () => { const a = ZERO12(); a[0] = 1n; return a; };
const fromFp = (c) => { const a = ZERO12(); a[0] = fp(c); return a; };
const add12 = (a, b) => a.map((v, i) => fp(v + b[i]));
const sub12 = (a, b) => a.map((v, i) => fp(v - b[i]));
const eq12 = (a, b) => a.every((v, i) => v === b[i]);
function mul12(a, b) {
  const t = new Array(23).fill(0n);
  for (let i = 0; i < 12; i++) { if (a[i] === 0n) continue; for (let j = 0; j < 12; j++) t[i + j] += a[i] * b[j]; }
  for (let i = 22; i >= 12; i--) { const c = t[i] % P; t[i - 6] += 18n * c; t[i - 12] -= 82n * c; }  // w^12 = 18 w^6 - 82
  return t.slice(0, 12).map(fp);
}
function inv12(a) {  // solve (multiplication-by-a matrix) x = 1 by Gauss-Jordan over Fp
  const m = [];
  let col = a.slice();
  const wpow = ZERO12(); wpow[1] = 1n;
  const cols = [];
  for (let j = 0; j < 12; j++) { cols.push(col); col = mul12(col, wpow); }
  for (let i = 0; i < 12; i++) { m.push(cols.map((c) => c[i])); m[i].push(i === 0 ? 1n : 0n); }
  for (let c = 0; c < 12; c++) {
    let piv = c;
    while (piv < 12 && m[piv][c] === 0n) piv++;
    if (piv === 12) throw new Error('inverse of zero in Fp12');
    [m[c], m[piv]] = [m[piv], m[c]];
    const s = fpInv(m[c][c]);
    m[c] = m[c].map((v) => fp(v * s));
    for (let r = 0; r < 12; r++) {
      if (r === c || m[r][c] === 0n) continue;
      const f = m[r][c];
      m[r] = m[r].map((v, k) => fp(v - f * m[c][k]));
    }
  }
  return m.map((row) => row[12]);
}
function pow12(a, e) {
  let r = ONE12();
  for (let i = e.toString(2).length - 1; i >= 0; i--) { r = mul12(r, r); if ((e >> BigInt(i)) & 1n) r = mul12(r, a); }
  return r;
}

// ---- E(Fp12): y^2 = x^3 + 3 ---------------------------------------------------------------------------------------------
// chord / tangent through a and b: returns [a + b, line evaluated at t]; the vertical line when a = -b
function step(a, b, t) {
  let m;
  if (!eq12(a[0], b[0])) m = mul12(sub12(b[1], a[1]), inv12(sub12(b[0], a[0])));
  else if (eq12(a[1], b[1])) m = mul12(mul12(fromFp(3n), mul12(a[0], a[0])), inv12(add12(a[1], a[1])));
  else return [null, sub12(t[0], a[0])];
  const line = sub12(mul12(m, sub12(t[0], a[0])), sub12(t[1], a[1]));
  const x = sub12(sub12(mul12(m, m), a[0]), b[0]);
  return [[x, sub12(mul12(m, sub12(a[0], x)), a[1])], line];
}
// G2 point ((x0, x1), (y0, y1)) over Fp2 = Fp[u]/(u^2 + 1), on the twist y^2 = x^3 + 3/(9 + u)  ->  E(Fp12)
function untwist(q) {
  const x = ZERO12(), y = ZERO12();
  x[2] = fp(q[0][0] - 9n * q[0][1]); x[8] = fp(q[0][1]);   // (x0 + x1 u) w^2 with u = w^6 - 9
  y[3] = fp(q[1][0] - 9n * q[1][1]); y[9] = fp(q[1][1]);   // (y0 + y1 u) w^3
  return [x, y];
}
function onCurve12(pt) { return eq12(mul12(pt[1], pt[1]), add12(mul12(mul12(pt[0], pt[0]), pt[0]), fromFp(3n))); }

function miller(p1, q2) {  // f_{6x+2, Q}(P) l_{[6x+2]Q, pi(Q)}(P) l_{.., -pi^2(Q)}(P), before the final exponentiation
  if (p1 === null || q2 === null) return ONE12();
  const q = untwist(q2), t = [fromFp(p1[0]), fromFp(p1[1])];
  if (!onCurve12(q) || !g1OnCurve(p1)) throw new Error('point not on its curve');
  let f = ONE12(), r = q, l;
  for (let i = ATE.toString(2).length - 2; i >= 0; i--) {
    [r, l] = step(r, r, t);
    f = mul12(mul12(f, f), l);
    if ((ATE >> BigInt(i)) & 1n) { [r, l] = step(r, q, t); f = mul12(f, l); }
  }
  const q1 = [pow12(q[0], P), pow12(q[1], P)];
  const nq2 = [pow12(q1[0], P), sub12(ZERO12(), pow12(q1[1], P))];
  [r, l] = step(r, q1, t); f = mul12(f, l);
  [r, l] = step(r, nq2, t); f = mul12(f, l);
  return f;
}
const FINAL = (P ** 12n - 1n) / R;
const finalExp = (f) => pow12(f, FINAL);
const pairing = (p1, q2) => finalExp(miller(p1, q2));

// Groth16: e(A, B) = e(alpha, beta) e(sum_i x_i IC_i, gamma) e(C, delta), as ONE product: e(-A, B) e(alpha, beta) ... = 1
function groth16Verify(vk, pub, proof) {
  if (pub.length + 1 !== vk.ic.length) return false;
  let acc = vk.ic[0];
  for (let i = 0; i < pub.length; i++) acc = g1Add(acc, g1Mul(vk.ic[i + 1], pub[i] % R));
  const negA = [proof.a[0], fp(-proof.a[1])];
  let f = miller(negA, proof.b);
  f = mul12(f, miller(vk.alpha, vk.beta));
  f = mul12(f, miller(acc, vk.gamma));
  f = mul12(f, miller(proof.c, vk.delta));
  return eq12(finalExp(f), ONE12());
}

// ---- driver ---------------------------------------------------------------------------------------------------------------
const big = (v) => {  // decimal strings -> BigInt, through arrays and objects
  if (Array.isArray(v)) return v.map(big);
  if (v !== null && typeof v === 'object') { const o = {}; for (const k of Object.keys(v)) o[k] = big(v[k]); return o; }
  return BigInt(v);
};
const G1 = [1n, 2n];
const G2 = [[10857046999023057135944570762232829481370756359578518086990519993285655852781n,
             11559732032986387107991004021392285783925812861821192530917403151452391805634n],
            [8495653923123431417604973247489272438418190587263600148770280649306958101930n,
             4082367875863433681332203403145435568316851327593401208105741076214120093531n]];
// `node bn254_pairing_second.js --snarkjs verification_key.json public.json proof.json`: the three files of
// `snarkjs groth16 verify` (decimal strings, projective points with z = 1, Fq2 as [c0, c1]; owshen_amd/snarkjs_json.py writes
// them) through the verifier above.  Prints OK / INVALID, exit code 0 / 1, like the command it stands in for.
function snarkjsPoint1(p) { if (BigInt(p[2]) !== 1n) throw new Error('G1 point not normalised (z != 1)'); return [BigInt(p[0]), BigInt(p[1])]; }
function snarkjsPoint2(p) {
  if (BigInt(p[2][0]) !== 1n || BigInt(p[2][1]) !== 0n) throw new Error('G2 point not normalised (z != 1)');
  return [[BigInt(p[0][0]), BigInt(p[0][1])], [BigInt(p[1][0]), BigInt(p[1][1])]];
}
if (process.argv[2] === '--snarkjs') {
  const fs = require('fs');
  const [vkj, pubj, prj] = process.argv.slice(3, 6).map((f) => JSON.parse(fs.readFileSync(f, 'utf8')));
  let ok = false;
  try {
    if (vkj.protocol !== 'groth16' || prj.protocol !== 'groth16') throw new Error('protocol is not groth16');
    if (vkj.curve !== 'bn128' || prj.curve !== 'bn128') throw new Error('curve is not bn128');
    if (vkj.nPublic !== pubj.length || vkj.IC.length !== pubj.length + 1) throw new Error('public input count');
    const vk = { alpha: snarkjsPoint1(vkj.vk_alpha_1), beta: snarkjsPoint2(vkj.vk_beta_2), gamma: snarkjsPoint2(vkj.vk_gamma_2),
                 delta: snarkjsPoint2(vkj.vk_delta_2), ic: vkj.IC.map(snarkjsPoint1) };
    const pub = pubj.map(BigInt);
    if (pub.some((x) => x < 0n || x >= R)) throw new Error('public input not below r');   // snarkjs: "Public input is not valid"
    ok = groth16Verify(vk, pub, { a: snarkjsPoint1(prj.pi_a), b: snarkjsPoint2(prj.pi_b), c: snarkjsPoint1(prj.pi_c) });
  } catch (err) { process.stderr.write(String(err) + '\n'); ok = false; }
  process.stdout.write(ok ? 'OK\n' : 'INVALID\n');
  process.exit(ok ? 0 : 1);
}
let text = '';
process.stdin.on('data', (d) => { text += d; });
process.stdin.on('end', () => {
  const req = text.trim() ? JSON.parse(text) : {};
  const out = {};
  const e = pairing(G1, G2);
  out.e_g1_g2 = e.map(String);
  out.e_not_one = !eq12(e, ONE12());
  out.e_order_r = eq12(pow12(e, R), ONE12());
  // bilinearity with honest scalar multiplications on both sides: e(5 G1, G2) = e(G1, G2)^5
  out.bilinear_g1 = eq12(pairing(g1Mul(G1, 5n), G2), pow12(e, 5n));
  out.proofs = (req.proofs || []).map((pr) => {
    try { return groth16Verify(big(req.vk), big(pr.public), big(pr.proof)); } catch (err) { return false; }  // (a point off its curve)
  });
  process.stdout.write(JSON.stringify(out));
});
