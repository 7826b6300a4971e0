/*
 * flx_mesh.hip — SURVEY.md 8f N2: OBJ / MTL import, BVH build and flattening in native code (host only, no GPU).
 *
 * What the reference does in JavaScript for every imported object — Scene.importMtl / importObj (modules/scene.js:330-487),
 * generateBVH (:62-154), updateBoundings (:157-187) and its share of generateArraysFromGraph (:190-316) — with the same
 * numerical recipe, because the ORDER and the BITS of what comes out are the input of the GPU hot path: the arrays of a
 * flx_mesh equal the block the JavaScript host layer (js/scene.js, itself pinned to the reference's own output by
 * tests/golden/ref_*.json) emits for the same object.  JavaScript numbers are doubles and typed arrays round once to
 * float: kept (double arithmetic, one rounding at every Float32Array store, no contraction: -ffp-contract=off).
 * The reference's quirks that decide bits are kept too and cited where they are.
 */
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/flexlight_hip.h"

namespace {

constexpr int LEAF_MAX = 4;                          /* scene.js:6 */
constexpr double NODE_BIAS = 0.00152587890625;       /* scene.js:159 (100 * 2^-16) */
const double EPS = std::pow(2.0, -32);               /* math.js:8 */

/* ---- the reference's Math extensions (modules/math.js:8-54) ---- */
double jsRound(double x) {                           /* Math.round: ties up, and -0 for x in [-0.5, -0] */
  const double r = std::floor(x + 0.5);
  return (r == 0.0 && std::signbit(x)) ? -0.0 : r;
}
double snap(double x) {                              /* math.js:10 */
  const double frac = std::fmod(std::fabs(x), 1.0);
  return (frac < EPS || frac > 1.0 - EPS) ? jsRound(x) : x;
}
struct V3 { double x, y, z; };
V3 sub(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
V3 cross(V3 a, V3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
V3 unit(V3 a) {                                      /* math.js:49-54 */
  const double len = snap(std::sqrt(((0.0 + a.x * a.x) + a.y * a.y) + a.z * a.z));
  if (snap(len) < EPS) return { 0.0, 0.0, 0.0 };
  return { snap(a.x / len), snap(a.y / len), snap(a.z / len) };
}

/* ---- a primitive: 1 (Triangle) or 2 (Plane) triangles; live typed arrays and the two flat buffers built from them by
 * flatten(), which only the setters call (scene.js:628-730) ---- */
struct Prim {
  int length = 1;
  float vertices[18] = {}, normals[18] = {}, uvs[12] = {};
  float textureNums[3] = { -1.0f, -1.0f, -1.0f }, albedo[3] = { 1.0f, 1.0f, 1.0f }, rme[3] = { 1.0f, 0.0f, 0.0f }, tpo[3] = { 0.0f, 0.0f, 1.0f };
  uint32_t transform = 0;
  float geometry[24] = {}, attributes[56] = {};
  double bounding[6] = {};

  void flatten() {
    for (int t = 0; t < length; t++) {
      float *g = geometry + t * 12, *a = attributes + t * 28;
      std::memcpy(g, vertices + t * 9, 9 * sizeof(float));
      g[9] = (float)transform; g[10] = 2.0f; g[11] = 0.0f;
      std::memcpy(a, normals + t * 9, 9 * sizeof(float));
      std::memcpy(a + 9, uvs + t * 6, 6 * sizeof(float));
      std::memcpy(a + 15, textureNums, 3 * sizeof(float));
      std::memcpy(a + 18, albedo, 3 * sizeof(float));
      std::memcpy(a + 21, rme, 3 * sizeof(float));
      std::memcpy(a + 24, tpo, 3 * sizeof(float));
      a[27] = 0.0f;
    }
  }
  void init(int len, const V3 *corners, int nCorners, V3 normal, const float *uv) {
    length = len;
    for (int i = 0; i < nCorners; i++) { vertices[3 * i] = (float)corners[i].x; vertices[3 * i + 1] = (float)corners[i].y; vertices[3 * i + 2] = (float)corners[i].z; }
    for (int i = 0; i < len * 3; i++) { normals[3 * i] = (float)normal.x; normals[3 * i + 1] = (float)normal.y; normals[3 * i + 2] = (float)normal.z; }
    std::memcpy(uvs, uv, (size_t)len * 6 * sizeof(float));
    flatten();
  }
};

/* ---- a node of the tree generateBVH builds: a group of nodes, or a leaf group of primitives ---- */
struct Node {
  std::vector<std::unique_ptr<Node>> children;       /* empty for a leaf item */
  int prim = -1;                                     /* index into the mesh's primitives when this node IS a primitive */
  double bounding[6] = {};
};

struct Material { bool hasColor = false, hasEmissive = false, hasMetal = false, hasIor = false; double color[3] = {}, emissiveness = 0, metallicity = 0, ior = 1; };

}  // namespace

struct flx_mesh {
  std::vector<Prim> prims;
  std::unique_ptr<Node> root;
  double relativePosition[3] = { 0, 0, 0 };
  uint32_t transform = 0;
  uint32_t entries = 0, triangles = 0;
  std::string error;
};

namespace {

/* scene.js:157-187 */
void updateBoundings(flx_mesh &m, Node &n) {
  if (n.prim >= 0) {
    Prim &p = m.prims[(size_t)n.prim];
    const float *v = p.vertices;
    double box[6] = { v[0], v[0], v[1], v[1], v[2], v[2] };
    for (int i = 3; i < p.length * 9; i++) {
      const int a = (i % 3) * 2;
      box[a] = std::fmin(box[a], (double)v[i]);
      box[a + 1] = std::fmax(box[a + 1], (double)v[i]);
    }
    std::memcpy(p.bounding, box, sizeof box);
    std::memcpy(n.bounding, box, sizeof box);
    return;
  }
  double box[6];
  updateBoundings(m, *n.children[0]);
  std::memcpy(box, n.children[0]->bounding, sizeof box);
  for (size_t i = 1; i < n.children.size(); i++) {
    updateBoundings(m, *n.children[i]);
    const double *b = n.children[i]->bounding;
    for (int k = 0; k < 6; k++) box[k] = (k % 2 == 0) ? std::fmin(box[k], b[k] - NODE_BIAS) : std::fmax(box[k], b[k] + NODE_BIAS);
  }
  std::memcpy(n.bounding, box, sizeof box);
}

bool fitsInBound(const double *bound, const double *b) {          /* scene.js:56-59 */
  return bound[0] <= b[0] && bound[2] <= b[2] && bound[4] <= b[4] && bound[1] >= b[1] && bound[3] >= b[3] && bound[5] >= b[5];
}

/* scene.js:62-154: split at the box centre on the axis with the fewest straddlers (the LAST axis among equals: ">="), three
 * buckets in the order below / above / straddling, leaves of <= 4, depth <= log2(n) + 8 */
std::unique_ptr<Node> split(flx_mesh &m, std::unique_ptr<Node> objs, int depth, double maxDepth) {
  if ((int)objs->children.size() <= LEAF_MAX || (double)depth > maxDepth) return objs;
  const double *bb = objs->bounding;
  const double centre[3] = { (bb[0] + bb[1]) / 2.0, (bb[2] + bb[3]) / 2.0, (bb[4] + bb[5]) / 2.0 };
  const double minWidth = 1.0 / 256.0;
  int axis = 0;
  double fewest = INFINITY;
  for (int a = 0; a < 3; a++) {
    double upper[6], lower[6];
    std::memcpy(upper, bb, sizeof upper); std::memcpy(lower, bb, sizeof lower);
    upper[a * 2] = centre[a];
    lower[a * 2 + 1] = centre[a];
    const double room = std::fmin(upper[a * 2 + 1] - centre[a], centre[a] - lower[a * 2]);
    double n = 0;
    for (auto &c : objs->children) if (!fitsInBound(upper, c->bounding) && !fitsInBound(lower, c->bounding)) n += 1;
    if (fewest >= n && room > minWidth) { axis = a; fewest = n; }
  }
  if (fewest == INFINITY) return objs;                /* no axis splits it: the subtree stays a flat list (modules/scene.js:128-132) */
  double b0[6], b1[6];
  std::memcpy(b0, bb, sizeof b0); std::memcpy(b1, bb, sizeof b1);
  b0[axis * 2] = centre[axis];
  b1[axis * 2 + 1] = centre[axis];
  std::vector<std::unique_ptr<Node>> buckets[3];
  for (auto &c : objs->children) {
    if (fitsInBound(b0, c->bounding)) buckets[0].push_back(std::move(c));
    else if (fitsInBound(b1, c->bounding)) buckets[1].push_back(std::move(c));
    else buckets[2].push_back(std::move(c));
  }
  std::unique_ptr<Node> parent(new Node());
  for (int k = 0; k < 3; k++) {
    if (buckets[k].empty()) continue;
    std::unique_ptr<Node> node(new Node());
    node->children = std::move(buckets[k]);
    updateBoundings(m, *node);
    parent->children.push_back(split(m, std::move(node), depth + 1, maxDepth));
  }
  return parent;
}

/* ---- text: lines at \r\n | \r | \n, words at white space or a literal '+' (the reference's class, scene.js:339) ---- */
struct Words { std::vector<std::string> w; };
bool isSep(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v' || c == '+'; }
template <class F> void forEachLine(const char *text, size_t len, F f) {
  size_t i = 0;
  while (i <= len) {
    size_t j = i;
    while (j < len && text[j] != '\n' && text[j] != '\r') j++;
    Words ws;
    size_t k = i;
    while (k < j) {
      while (k < j && isSep(text[k])) k++;
      size_t e = k;
      while (e < j && !isSep(text[e])) e++;
      if (e > k) ws.w.emplace_back(text + k, e - k);
      k = e;
    }
    f(ws);
    if (j >= len) break;
    i = (text[j] == '\r' && j + 1 < len && text[j + 1] == '\n') ? j + 2 : j + 1;
  }
}
/* Number(string): the empty string is 0, garbage is NaN */
double jsNumber(const std::string &s) {
  if (s.empty()) return 0.0;
  char *end = nullptr;
  const double v = std::strtod(s.c_str(), &end);
  if (end == s.c_str() || *end != '\0') return NAN;
  return v;
}
double word(const Words &ws, size_t i) { return i < ws.w.size() ? jsNumber(ws.w[i]) : NAN; }   /* Number(undefined) */

/* scene.js:438-487 */
std::map<std::string, Material> importMtl(const char *text, size_t len) {
  std::map<std::string, Material> mats;
  Material *cur = nullptr;
  forEachLine(text, len, [&](const Words &ws) {
    if (ws.w.empty()) return;
    const std::string &k = ws.w[0];
    if (k == "newmtl") { cur = &mats[ws.w.size() > 1 ? ws.w[1] : std::string("undefined")]; *cur = Material(); }
    else if (!cur) return;
    else if (k == "Ka") { cur->hasColor = true; for (int i = 0; i < 3; i++) cur->color[i] = snap(word(ws, 1 + i) * 255.0); }
    else if (k == "Ke") {
      const double e = std::fmax(std::fmax(word(ws, 1), word(ws, 2)), word(ws, 3));
      if (e > 0) { cur->hasEmissive = true; cur->emissiveness = e * 4.0; cur->hasColor = true; for (int i = 0; i < 3; i++) cur->color[i] = snap(word(ws, 1 + i) * (255.0 / e)); }
    }
    else if (k == "Ns") { cur->hasMetal = true; cur->metallicity = word(ws, 1) / 1000.0; }
    else if (k == "Ni") { cur->hasIor = true; cur->ior = word(ws, 1); }
  });
  return mats;
}

/* scene.js:330-436 */
void importObj(flx_mesh &m, const char *text, size_t len, const std::map<std::string, Material> &mats) {
  std::vector<V3> v, vn;
  std::vector<std::pair<double, double>> vt;
  const Material *material = nullptr;
  forEachLine(text, len, [&](const Words &ws) {
    if (ws.w.empty()) return;
    const std::string &k = ws.w[0];
    if (k == "v") v.push_back({ word(ws, 1), word(ws, 2), word(ws, 3) });
    else if (k == "vt") vt.push_back({ word(ws, 1), word(ws, 2) });
    else if (k == "vn") vn.push_back({ word(ws, 1), word(ws, 2), word(ws, 3) });
    else if (k == "usemtl") {
      auto it = ws.w.size() > 1 ? mats.find(ws.w[1]) : mats.end();
      if (it != mats.end()) material = &it->second;              /* an unknown name keeps the current material */
    } else if (k == "f") {
      /* corner = up to three numbers v/vt/vn; negative ones count from the end OF THE VERTEX LIST, all three (scene.js:356) */
      std::vector<std::array<double, 3>> corners;
      for (size_t c = 1; c < ws.w.size(); c++) {
        std::array<double, 3> idx = { NAN, NAN, NAN };
        const std::string &s = ws.w[c];
        size_t a = 0; int field = 0;
        while (field < 3) {
          size_t b = s.find('/', a);
          const std::string part = s.substr(a, b == std::string::npos ? std::string::npos : b - a);
          double n = jsNumber(part);
          if (n < 0) n = (double)v.size() + n + 1.0;
          idx[(size_t)field++] = n;
          if (b == std::string::npos) break;
          a = b + 1;
        }
        corners.push_back(idx);
      }
      if (corners.size() < 3) return;
      auto vertexAt = [&](double n) -> V3 { const long long i = (long long)n - 1; return (n == n && i >= 0 && (size_t)i < v.size()) ? v[(size_t)i] : V3{ NAN, NAN, NAN }; };
      Prim p;
      int order[6]; int nOrder;
      static const float uvPlane[12] = { 0, 0, 0, 1, 1, 1, 1, 1, 1, 0, 0, 0 }, uvTri[6] = { 0, 0, 0, 1, 1, 1 };
      if (corners.size() == 4) {
        const V3 c0 = vertexAt(corners[3][0]), c1 = vertexAt(corners[2][0]), c2 = vertexAt(corners[1][0]), c3 = vertexAt(corners[0][0]);
        const V3 six[6] = { c0, c1, c2, c2, c3, c0 };
        p.init(2, six, 6, unit(cross(sub(c0, c2), sub(c0, c1))), uvPlane);               /* scene.js:747-751 */
        const int o[6] = { 3, 2, 1, 1, 0, 3 }; std::memcpy(order, o, sizeof o); nOrder = 6;
      } else {
        const V3 a = vertexAt(corners[2][0]), b = vertexAt(corners[1][0]), c = vertexAt(corners[0][0]);
        const V3 three[3] = { a, b, c };
        p.init(1, three, 3, unit(cross(sub(a, c), sub(a, b))), uvTri);                    /* scene.js:753-757 */
        const int o[3] = { 2, 1, 0 }; std::memcpy(order, o, sizeof o); nOrder = 3;
      }
      /* uvs / normals go into the live arrays; they reach the flat buffers only if a setter runs later (scene.js:381-400) */
      for (int i = 0; i < nOrder; i++) {
        const std::array<double, 3> &c = corners[(size_t)order[i]];
        const long long ti = (long long)c[1] - 1, ni = (long long)c[2] - 1;
        if (c[1] == c[1] && ti >= 0 && (size_t)ti < vt.size()) { p.uvs[2 * i] = (float)vt[(size_t)ti].first; p.uvs[2 * i + 1] = (float)vt[(size_t)ti].second; }
        if (c[2] == c[2] && ni >= 0 && (size_t)ni < vn.size()) { p.normals[3 * i] = (float)vn[(size_t)ni].x; p.normals[3 * i + 1] = (float)vn[(size_t)ni].y; p.normals[3 * i + 2] = (float)vn[(size_t)ni].z; }
      }
      if (material) {
        const Material &mt = *material;
        for (int i = 0; i < 3; i++) p.albedo[i] = (float)((mt.hasColor ? mt.color[i] : 255.0) / 255.0);
        p.rme[2] = (float)(mt.hasEmissive ? mt.emissiveness : 0.0);
        p.rme[1] = (float)(mt.hasMetal ? mt.metallicity : 0.0);
        p.rme[0] = 1.0f;                                   /* materials carry no roughness: default 1 */
        p.tpo[0] = 0.0f;
        p.tpo[2] = (float)(mt.hasIor ? mt.ior : 1.0);
        p.flatten();
      }
      m.prims.push_back(p);
    }
  });
  /* generateBVH(items) then updateBoundings(items) (scene.js:433-435) */
  std::unique_ptr<Node> top(new Node());
  for (size_t i = 0; i < m.prims.size(); i++) { std::unique_ptr<Node> leaf(new Node()); leaf->prim = (int)i; top->children.push_back(std::move(leaf)); }
  if (top->children.empty()) { m.root = std::move(top); return; }
  updateBoundings(m, *top);
  const double maxDepth = std::log2((double)top->children.size()) + 8.0;
  m.root = split(m, std::move(top), 0, maxDepth);
  updateBoundings(m, *m.root);
}

void measure(const flx_mesh &m, const Node &n, uint32_t &entries, uint32_t &triangles) {
  if (n.prim >= 0) { entries += (uint32_t)m.prims[(size_t)n.prim].length; triangles += (uint32_t)m.prims[(size_t)n.prim].length; return; }
  if (n.children.empty()) return;
  entries++;
  for (auto &c : n.children) measure(m, *c, entries, triangles);
}

/* scene.js:190-316 for this subtree: depth-first, a group's entry first (tight box of what it holds, entries to skip on a miss) */
void emit(const flx_mesh &m, const Node &n, float *geometry, float *attributes, int32_t *ids, uint32_t &at, uint32_t &tri, float box[6]) {
  if (n.prim >= 0) {
    const Prim &p = m.prims[(size_t)n.prim];
    std::memcpy(geometry + (size_t)at * 12, p.geometry, (size_t)p.length * 12 * sizeof(float));
    std::memcpy(attributes + (size_t)at * 28, p.attributes, (size_t)p.length * 28 * sizeof(float));
    for (int i = 0; i < p.length; i++) ids[tri++] = (int32_t)at++;
    const float *v = p.vertices;
    box[0] = box[3] = v[0]; box[1] = box[4] = v[1]; box[2] = box[5] = v[2];
    for (int i = 3; i < p.length * 9; i += 3)
      for (int k = 0; k < 3; k++) { box[k] = std::fmin(box[k], v[i + k]); box[k + 3] = std::fmax(box[k + 3], v[i + k]); }
    return;
  }
  const uint32_t self = at++;
  emit(m, *n.children[0], geometry, attributes, ids, at, tri, box);
  for (size_t i = 1; i < n.children.size(); i++) {
    float b[6];
    emit(m, *n.children[i], geometry, attributes, ids, at, tri, b);
    for (int k = 0; k < 3; k++) { box[k] = std::fmin(box[k], b[k]); box[k + 3] = std::fmax(box[k + 3], b[k + 3]); }
  }
  float *g = geometry + (size_t)self * 12;
  for (int k = 0; k < 6; k++) g[k] = box[k];
  g[6] = (float)(at - self - 1);
  g[7] = g[8] = 0.0f;
  g[9] = (float)m.transform;
  g[10] = 1.0f; g[11] = 0.0f;
  std::memset(attributes + (size_t)self * 28, 0, 28 * sizeof(float));
}

}  // namespace

/* ---- the per-frame transform arrays (Transform.buildWGL2Arrays, scene.js:500-521) in native code -------------
 * The inverse is the reference's Moore-Penrose inverse through a Gram-Schmidt QR of A^T A with its `stabilize` snapping
 * (math.js:56-101); every operation in the order the JavaScript does it, so the arrays are the same floats. */
namespace {
typedef std::vector<std::vector<double>> Mat;
std::vector<double> scaleVec(const std::vector<double> &v, double s) { std::vector<double> r(v.size()); for (size_t i = 0; i < v.size(); i++) r[i] = snap(v[i] * s); return r; }
std::vector<double> addVec(const std::vector<double> &a, const std::vector<double> &b) { std::vector<double> r(a.size()); for (size_t i = 0; i < a.size(); i++) r[i] = a[i] + b[i]; return r; }
double dotN(const std::vector<double> &a, const std::vector<double> &b) {              /* math.js:41: snapped products, summed from 0, snapped */
  double p = 0.0;
  for (size_t i = 0; i < a.size(); i++) p = p + snap(a[i] * b[i]);
  return snap(p);
}
std::vector<double> unitN(const std::vector<double> &a) {                                /* math.js:49-54 */
  double q = 0.0;
  for (double c : a) q = q + c * c;
  const double len = snap(std::sqrt(q));
  std::vector<double> r(a.size());
  for (size_t i = 0; i < a.size(); i++) r[i] = (snap(len) < EPS) ? 0.0 : snap(a[i] / len);
  return r;
}
Mat transposeM(const Mat &A) { Mat T(A[0].size(), std::vector<double>(A.size())); for (size_t i = 0; i < A.size(); i++) for (size_t j = 0; j < A[0].size(); j++) T[j][i] = A[i][j]; return T; }
Mat matMul(const Mat &A, const Mat &B) {                                                /* math.js:15-19 */
  const Mat BT = transposeM(B);
  Mat R(A.size(), std::vector<double>(BT.size()));
  for (size_t i = 0; i < A.size(); i++) for (size_t j = 0; j < BT.size(); j++) R[i][j] = dotN(A[i], BT[j]);
  return R;
}
Mat gramSchmidt(const Mat &A) {                                                         /* math.js:62-71, over the rows */
  Mat B;
  for (const auto &row : A) {
    std::vector<double> proj(A[0].size(), 0.0);
    for (const auto &c : B) proj = addVec(proj, scaleVec(c, dotN(c, row) / dotN(c, c)));
    B.push_back(addVec(row, scaleVec(proj, -1.0)));
  }
  return B;
}
Mat pseudoInverse(const Mat &A, int depth = 0) {                                        /* math.js:78-101 */
  const Mat AT = transposeM(A);
  const Mat M = matMul(AT, A);
  Mat QT = gramSchmidt(transposeM(M));
  for (auto &r : QT) r = unitN(r);
  const Mat R = matMul(QT, M);
  const size_t n = R.size();
  Mat Rinv(n);
  for (size_t ii = n; ii-- > 0;) {
    Rinv[ii].assign(n, 0.0); Rinv[ii][ii] = 1.0;
    for (size_t j = n; j-- > ii + 1;) Rinv[ii] = addVec(Rinv[ii], scaleVec(Rinv[j], -R[ii][j] / R[j][j]));
  }
  for (size_t i = 0; i < n; i++) Rinv[i] = scaleVec(Rinv[i], 1.0 / R[i][i]);
  if (Rinv[0][0] != Rinv[0][0] && depth == 0) return transposeM(pseudoInverse(AT, 1));
  return matMul(matMul(Rinv, QT), AT);                /* transpose(Q) = QT */
}
}  // namespace

extern "C" flx_status flx_transforms_pack(uint32_t n_transforms, const double *matrices, const double *positions, float *rotation, float *shift) {
  if (!matrices || !positions || !rotation || !shift) return FLX_ERR_INVALID;
  for (uint32_t t = 0; t < n_transforms; t++) {
    Mat m(3, std::vector<double>(3));
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) m[(size_t)r][(size_t)c] = matrices[(size_t)t * 9 + (size_t)r * 3 + (size_t)c];
    const Mat inv = pseudoInverse(m);
    float *rot = rotation + (size_t)t * 24, *sh = shift + (size_t)t * 8;
    std::memset(rot, 0, 24 * sizeof(float)); std::memset(sh, 0, 8 * sizeof(float));
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {            /* JS rows land in GLSL columns (scene.js:510-517) */
      rot[4 * r + c] = (float)m[(size_t)r][(size_t)c];
      rot[12 + 4 * r + c] = (float)inv[(size_t)r][(size_t)c];
    }
    for (int k = 0; k < 3; k++) { sh[k] = (float)positions[(size_t)t * 3 + (size_t)k]; sh[4 + k] = (float)snap(positions[(size_t)t * 3 + (size_t)k] * -1.0); }
  }
  return FLX_OK;
}

extern "C" flx_status flx_mesh_import_obj(const char *obj_text, size_t obj_len, const char *mtl_text, size_t mtl_len, flx_mesh **out) {
  if (!obj_text || !out) return FLX_ERR_INVALID;
  std::unique_ptr<flx_mesh> m(new flx_mesh());
  std::map<std::string, Material> mats;
  if (mtl_text && mtl_len) mats = importMtl(mtl_text, mtl_len);
  importObj(*m, obj_text, obj_len, mats);
  if (m->root) measure(*m, *m->root, m->entries, m->triangles);
  *out = m.release();
  return FLX_OK;
}

extern "C" void flx_mesh_destroy(flx_mesh *m) { delete m; }
extern "C" uint32_t flx_mesh_entry_count(const flx_mesh *m) { return m ? m->entries : 0u; }
extern "C" uint32_t flx_mesh_triangle_count(const flx_mesh *m) { return m ? m->triangles : 0u; }

extern "C" flx_status flx_mesh_set_transform(flx_mesh *m, uint32_t transform_number) {      /* scene.js:774-779: every node and primitive */
  if (!m) return FLX_ERR_INVALID;
  m->transform = transform_number;
  for (Prim &p : m->prims) { p.transform = transform_number; p.flatten(); }
  return FLX_OK;
}

extern "C" flx_status flx_mesh_move(flx_mesh *m, double x, double y, double z) {             /* scene.js:811-829: adds to the vertices */
  if (!m) return FLX_ERR_INVALID;
  const double d[3] = { x, y, z };
  m->relativePosition[0] = x; m->relativePosition[1] = y; m->relativePosition[2] = z;
  for (Prim &p : m->prims) { for (int i = 0; i < p.length * 9; i++) p.vertices[i] = (float)((double)p.vertices[i] + d[i % 3]); p.flatten(); }
  return FLX_OK;
}

extern "C" flx_status flx_mesh_scale(flx_mesh *m, double s) {                                 /* scene.js:831-839: about relativePosition */
  if (!m) return FLX_ERR_INVALID;
  const double *o = m->relativePosition;
  for (Prim &p : m->prims) { for (int i = 0; i < p.length * 9; i++) p.vertices[i] = (float)(((double)p.vertices[i] - o[i % 3]) * s + o[i % 3]); p.flatten(); }
  return FLX_OK;
}

extern "C" flx_status flx_mesh_set_material(flx_mesh *m, int field, const double *values) {   /* the broadcast setters, scene.js:781-808 */
  if (!m || !values) return FLX_ERR_INVALID;
  for (Prim &p : m->prims) {
    switch (field) {
      case FLX_MESH_COLOR: for (int i = 0; i < 3; i++) p.albedo[i] = (float)(values[i] / 255.0); break;      /* 0..255 in, /255 stored */
      case FLX_MESH_ROUGHNESS: p.rme[0] = (float)values[0]; break;
      case FLX_MESH_METALLICITY: p.rme[1] = (float)values[0]; break;
      case FLX_MESH_EMISSIVENESS: p.rme[2] = (float)values[0]; break;
      case FLX_MESH_TRANSLUCENCY: p.tpo[0] = (float)values[0]; break;
      case FLX_MESH_IOR: p.tpo[2] = (float)values[0]; break;
      case FLX_MESH_TEXTURE_NUMS: for (int i = 0; i < 3; i++) p.textureNums[i] = (float)values[i]; break;
      default: return FLX_ERR_INVALID;
    }
    p.flatten();
  }
  return FLX_OK;
}

extern "C" flx_status flx_mesh_bounding(flx_mesh *m, double box[6]) {      /* Scene.updateBoundings of the object: [xmin,xmax,ymin,ymax,zmin,zmax] */
  if (!m || !box) return FLX_ERR_INVALID;
  if (!m->root || m->entries == 0u) { for (int k = 0; k < 6; k++) box[k] = 0.0; return FLX_OK; }
  updateBoundings(*m, *m->root);
  std::memcpy(box, m->root->bounding, 6 * sizeof(double));
  return FLX_OK;
}

extern "C" flx_status flx_mesh_flatten(const flx_mesh *m, float *geometry, float *attributes, int32_t *ids, float minmax[6]) {
  if (!m || !geometry || !attributes || !ids) return FLX_ERR_INVALID;
  if (!m->root || m->entries == 0u) return FLX_OK;
  uint32_t at = 0, tri = 0;
  float box[6] = {};
  emit(*m, *m->root, geometry, attributes, ids, at, tri, box);
  if (minmax) std::memcpy(minmax, box, sizeof box);
  return FLX_OK;
}
