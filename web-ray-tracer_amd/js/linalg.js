'use strict';
/*
 * Small dense linear algebra for the scene host layer.  The arithmetic (operation order and the
 * "stabilize" snapping) is that of the reference's Math extensions (modules/math.js:8-101) because
 * the bits of plane normals and of the transform inverses end up in the arrays the GPU consumes,
 * and those arrays are compared with the reference's own output bit for bit (tests/test_js_host.py).
 * Unlike the reference nothing here is attached to the global Math object.
 */
const EPS = Math.pow(2, -32);                       // math.js:8

// math.js:10 — snap values within 2^-32 of an integer onto it
function snap (x) {
  const frac = Math.abs(x) % 1;
  return (frac < EPS || frac > 1 - EPS) ? Math.round(x) : x;
}

const scaleVec = (v, s) => v.map(e => snap(e * s));           // math.js:24,28
const hadamard = (a, b) => a.map((e, i) => snap(e * b[i]));   // math.js:29
const addVec = (a, b) => a.map((e, i) => e + b[i]);           // math.js:45
const subVec = (a, b) => a.map((e, i) => e - b[i]);           // math.js:47
const dot = (a, b) => snap(hadamard(a, b).reduce((p, c) => p + c, 0));   // math.js:41
const cross = (a, b) => [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]];
const norm = a => snap(Math.sqrt(a.reduce((p, c) => p + c ** 2, 0)));   // math.js:49
function unit (a) {                                           // math.js:51-54
  const len = norm(a);
  return a.map(e => (snap(len) < EPS ? 0 : snap(e / len)));
}
const transpose = A => A[0].map((_, j) => A.map(row => row[j]));
const scaleMat = (A, s) => A.map(row => row.map(e => e * s)); // math.js:25,33: matrices are NOT snapped
function matMul (A, B) {                                      // math.js:15-19
  const BT = transpose(B);
  return A.map(row => BT.map(col => dot(row, col)));
}
function identity (n) {
  const I = [];
  for (let i = 0; i < n; i++) { I.push(new Array(n).fill(0)); I[i][i] = 1; }
  return I;
}

// math.js:62-71 — classical Gram-Schmidt over the rows of A
function gramSchmidt (A) {
  const B = [];
  A.forEach(row => {
    const proj = B.reduce((p, c) => addVec(p, scaleVec(c, dot(c, row) / dot(c, c))), new Array(A[0].length).fill(0));
    B.push(addVec(row, scaleVec(proj, -1)));
  });
  return B;
}

// math.js:78-84
function qr (A) {
  const QT = gramSchmidt(transpose(A)).map(unit);
  return { Q: transpose(QT), R: matMul(QT, A) };
}

// math.js:86-101 — Moore-Penrose inverse through QR of A^T A, as Transform.buildWGL2Arrays uses it (scene.js:507)
function pseudoInverse (A) {
  const AT = transpose(A);
  const f = qr(matMul(AT, A));
  const n = f.R.length;
  const Rinv = new Array(n);
  for (let i = n - 1; i >= 0; i--) {
    Rinv[i] = f.R.map((_, j) => (i === j ? 1 : 0));
    for (let j = n - 1; j > i; j--) Rinv[i] = addVec(Rinv[i], scaleVec(Rinv[j], -f.R[i][j] / f.R[j][j]));
  }
  for (let i = 0; i < n; i++) Rinv[i] = scaleVec(Rinv[i], 1 / f.R[i][i]);
  if (Number.isNaN(Rinv[0][0])) return transpose(pseudoInverse(AT));
  return matMul(matMul(Rinv, transpose(f.Q)), AT);
}

module.exports = { EPS, snap, scaleVec, hadamard, addVec, subVec, dot, cross, norm, unit, transpose, scaleMat, matMul, identity, gramSchmidt, qr, pseudoInverse };
