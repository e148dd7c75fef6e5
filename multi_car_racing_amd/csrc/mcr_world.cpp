// mcr_world.cpp — the ONE b2World the reference keeps for the life of an env (multi_car_racing.py:138; _destroy :173-181; reset :341),
// reduced to what outlives an episode and reaches a result: the broadphase's dynamic AABB tree, whose node indices are Box2D's PROXY IDS.
// Host code, no GPU (include/mcr.h: mcr_world_*); used by the single-env facade (env.py), which synchronises every step anyway.
//
// Why it matters: the pair buffer of b2BroadPhase::UpdatePairs is sorted by (proxyIdA = min, proxyIdB = max) before b2ContactManager::AddPair
// sees it, contacts are pushed at the head of the world's list and b2ContactManager::Collide walks from the head — so the ids decide the order
// of the Begin callbacks of a step (FrictionDetector, mcr.py:80-123: which of two cars that reach a tile in the same step is its FIRST
// visitor) and which fixture of a pair is fixtureA.  In the first episode of a world the ids ascend in creation order; after reset() the
// new fixtures pop their ids off the tree's LIFO free list, in an order that depends on every InsertLeaf / RemoveLeaf / rotation since the
// world was made.  The batched envs (vec_env.py) define every episode as the first of a fresh world; the facade carries this tree.
//
// Restated from the published Box2D 2.3.x sources: b2DynamicTree.{h,cpp} (AllocateNode / FreeNode / InsertLeaf with the surface-area
// heuristic / RemoveLeaf / Balance), b2BroadPhase::{CreateProxy, DestroyProxy, MoveProxy}, b2Fixture::{CreateProxies, Synchronize},
// b2Body::SynchronizeFixtures, b2World::{DestroyBody, Solve}'s traversal orders, b2PolygonShape::ComputeAABB.
#include "../../include/mcr.h"
#include "mcr_common.h"
#include <vector>
#include <algorithm>
#include <cstring>

void mcr_build_shapes(McrShapes* S);   // mcr_host.cpp

namespace {

struct Box { float lx, ly, hx, hy; };
inline Box box_union(const Box& a, const Box& b) { Box c = {std::min(a.lx, b.lx), std::min(a.ly, b.ly), std::max(a.hx, b.hx), std::max(a.hy, b.hy)}; return c; }   // b2AABB::Combine
inline float box_perimeter(const Box& a) { const float wx = a.hx - a.lx, wy = a.hy - a.ly; return 2.0f * (wx + wy); }                                              // b2AABB::GetPerimeter
inline bool box_contains(const Box& a, const Box& b) { return a.lx <= b.lx && a.ly <= b.ly && b.hx <= a.hx && b.hy <= a.hy; }                                   // b2AABB::Contains
inline Box box_fatten(const Box& a) { Box f = {a.lx - 0.1f, a.ly - 0.1f, a.hx + 0.1f, a.hy + 0.1f}; return f; }                                                  // b2_aabbExtension

// b2DynamicTree: a pool of nodes (leaves = proxies, the rest internal), free nodes chained through `link`
class AabbTree {
 public:
  AabbTree() { grow(16); free_head_ = 0; }
  int create_proxy(const Box& tight) { const int id = take_node(); node_[id].box = box_fatten(tight); node_[id].height = 0; insert_leaf(id); return id; }
  void destroy_proxy(int id) { remove_leaf(id); give_node(id); }
  void move_proxy_to(int id, const Box& fat) { remove_leaf(id); node_[id].box = fat; insert_leaf(id); }      // (the caller decided to move and computed `fat` the way b2DynamicTree::MoveProxy does)
  const Box& fat_box(int id) const { return node_[id].box; }
  int node_count() const { return (int)node_.size(); }

 private:
  struct Node { Box box; int link /* parent, or the next free node */, kid1, kid2, height; };
  std::vector<Node> node_;
  int root_ = -1, free_head_ = -1;
  bool leaf(int i) const { return node_[i].kid1 == -1; }
  void grow(int to) {                                                 // new nodes are chained in ascending order (b2DynamicTree ctor / AllocateNode)
    const int from = (int)node_.size();
    node_.resize(to);
    for (int i = from; i < to; ++i) { node_[i].link = i + 1 < to ? i + 1 : -1; node_[i].height = -1; node_[i].kid1 = node_[i].kid2 = -1; }
  }
  int take_node() {
    if (free_head_ == -1) { const int cap = (int)node_.size(); grow(2 * cap); free_head_ = cap; }
    const int id = free_head_;
    free_head_ = node_[id].link;
    node_[id].link = -1; node_[id].kid1 = node_[id].kid2 = -1; node_[id].height = 0;
    return id;
  }
  void give_node(int id) { node_[id].link = free_head_; node_[id].height = -1; free_head_ = id; }
  void refit_upwards(int i, bool height_first) {
    while (i != -1) {
      i = balance(i);
      const int a = node_[i].kid1, b = node_[i].kid2;
      if (height_first) { node_[i].height = 1 + std::max(node_[a].height, node_[b].height); node_[i].box = box_union(node_[a].box, node_[b].box); }
      else { node_[i].box = box_union(node_[a].box, node_[b].box); node_[i].height = 1 + std::max(node_[a].height, node_[b].height); }
      i = node_[i].link;
    }
  }
  void insert_leaf(int leaf_id) {
    if (root_ == -1) { root_ = leaf_id; node_[leaf_id].link = -1; return; }
    const Box lb = node_[leaf_id].box;
    int at = root_;
    while (!leaf(at)) {                                               // descend by the surface-area heuristic
      const int k1 = node_[at].kid1, k2 = node_[at].kid2;
      const float area = box_perimeter(node_[at].box);
      const float merged = box_perimeter(box_union(node_[at].box, lb));
      const float cost_here = 2.0f * merged;
      const float inherit = 2.0f * (merged - area);
      float c1, c2;
      if (leaf(k1)) c1 = box_perimeter(box_union(lb, node_[k1].box)) + inherit;
      else { const float before = box_perimeter(node_[k1].box), after = box_perimeter(box_union(lb, node_[k1].box)); c1 = (after - before) + inherit; }
      if (leaf(k2)) c2 = box_perimeter(box_union(lb, node_[k2].box)) + inherit;
      else { const float before = box_perimeter(node_[k2].box), after = box_perimeter(box_union(lb, node_[k2].box)); c2 = after - before + inherit; }
      if (cost_here < c1 && cost_here < c2) break;
      at = c1 < c2 ? k1 : k2;
    }
    const int sibling = at, old_parent = node_[sibling].link, fresh = take_node();
    node_[fresh].link = old_parent;
    node_[fresh].box = box_union(lb, node_[sibling].box);
    node_[fresh].height = node_[sibling].height + 1;
    if (old_parent != -1) { if (node_[old_parent].kid1 == sibling) node_[old_parent].kid1 = fresh; else node_[old_parent].kid2 = fresh; }
    else root_ = fresh;
    node_[fresh].kid1 = sibling; node_[fresh].kid2 = leaf_id;
    node_[sibling].link = fresh; node_[leaf_id].link = fresh;
    refit_upwards(node_[leaf_id].link, true);
  }
  void remove_leaf(int leaf_id) {
    if (leaf_id == root_) { root_ = -1; return; }
    const int parent = node_[leaf_id].link, grand = node_[parent].link;
    const int sibling = node_[parent].kid1 == leaf_id ? node_[parent].kid2 : node_[parent].kid1;
    if (grand == -1) { root_ = sibling; node_[sibling].link = -1; give_node(parent); return; }
    if (node_[grand].kid1 == parent) node_[grand].kid1 = sibling; else node_[grand].kid2 = sibling;
    node_[sibling].link = grand;
    give_node(parent);
    refit_upwards(grand, false);
  }
  // one rotation if the sub-tree under `a` leans by more than one level; returns the sub-tree's new root
  int balance(int a) {
    if (leaf(a) || node_[a].height < 2) return a;
    const int b = node_[a].kid1, c = node_[a].kid2;
    const int lean = node_[c].height - node_[b].height;
    if (lean > 1) return rotate_up(a, c, b, false);
    if (lean < -1) return rotate_up(a, b, c, true);
    return a;
  }
  // `up` (a child of `a`) takes a's place; a keeps `keep` (its other child) and adopts the shallower child of `up`
  int rotate_up(int a, int up, int keep, bool up_is_first) {
    const int g1 = node_[up].kid1, g2 = node_[up].kid2;
    node_[up].kid1 = a; node_[up].link = node_[a].link; node_[a].link = up;
    const int above = node_[up].link;
    if (above != -1) { if (node_[above].kid1 == a) node_[above].kid1 = up; else node_[above].kid2 = up; }
    else root_ = up;
    const bool first_deeper = node_[g1].height > node_[g2].height;
    const int stays = first_deeper ? g1 : g2, moves = first_deeper ? g2 : g1;
    node_[up].kid2 = stays;
    if (up_is_first) node_[a].kid1 = moves; else node_[a].kid2 = moves;
    node_[moves].link = a;
    node_[a].box = box_union(node_[keep].box, node_[moves].box);
    node_[up].box = box_union(node_[a].box, node_[stays].box);
    node_[a].height = 1 + std::max(node_[keep].height, node_[moves].height);
    node_[up].height = 1 + std::max(node_[a].height, node_[stays].height);
    return up;
  }
};

inline const McrPoly& fixture_poly(const McrShapes& S, int f) { return f < 4 ? S.hull[f] : S.wheel; }
inline int fixture_body(int f) { return f < 4 ? 0 : f - 3; }
// b2PolygonShape::ComputeAABB
Box poly_box(const McrPoly& P, const Xf& xf) {
  V2 lo = xmul(xf, v2(P.vx[0], P.vy[0])), hi = lo;
  for (int i = 1; i < P.n; ++i) { const V2 v = xmul(xf, v2(P.vx[i], P.vy[i])); lo = v2(std::min(lo.x, v.x), std::min(lo.y, v.y)); hi = v2(std::max(hi.x, v.x), std::max(hi.y, v.y)); }
  Box b = {lo.x - B2_POLYGON_RADIUS, lo.y - B2_POLYGON_RADIUS, hi.x + B2_POLYGON_RADIUS, hi.y + B2_POLYGON_RADIUS};
  return b;
}

}  // namespace

struct mcr_world {
  int N;
  McrShapes S;
  AabbTree tree;
  std::vector<int> tile_id, fix_id;     // proxy ids of the live episode: [T], [N * 8] (fixture 0..3 hull polygons, 4..7 wheels)
  std::vector<Xf> xf;                   // [N * 5] body transforms the next step is entered with
};

extern "C" mcr_world* mcr_world_create(int num_agents) {
  if (num_agents < 1 || num_agents > MCR_MAX_AGENTS) return nullptr;
  mcr_world* w = new mcr_world();
  w->N = num_agents;
  mcr_build_shapes(&w->S);
  return w;
}
extern "C" void mcr_world_destroy(mcr_world* w) { delete w; }

// reset() on the world: _destroy (:173-181) — DestroyBody of every tile in road order, then per car the hull (b2World::DestroyBody walks the
// body's fixture list from its head, the fixture created LAST: polygons 3, 2, 1, 0) and its wheels in order (gym Car.destroy) —, then
// _create_track's tiles in track order (:318-327) and the cars by car id (:366-406), hull polygons then wheels (b2Fixture::CreateProxies at
// the spawn transform).  The ids go into the blob, where the contact pass looks them up.
extern "C" int mcr_world_reset(mcr_world* w, void* blob_io) {
  if (!w || !blob_io) return MCR_ERR_ARG;
  uint8_t* blob = (uint8_t*)blob_io;
  McrSlotHeader* H = (McrSlotHeader*)blob;
  const int T = H->T, N = w->N, F = 8 * N;
  if (T < 1 || T > MCR_TILE_CAP) return MCR_ERR_CAPACITY;
  for (int id : w->tile_id) w->tree.destroy_proxy(id);
  if (!w->fix_id.empty()) for (int c = 0; c < N; ++c) { for (int f = 3; f >= 0; --f) w->tree.destroy_proxy(w->fix_id[c * 8 + f]); for (int f = 4; f < 8; ++f) w->tree.destroy_proxy(w->fix_id[c * 8 + f]); }
  w->tile_id.assign(T, -1); w->fix_id.assign(F, -1); w->xf.assign((size_t)N * 5, Xf());
  const float* TA = (const float*)(blob + MCR_OFF_TAABB);               // the tile polygons' vertex boxes (the bodies sit at the origin, unrotated)
  for (int t = 0; t < T; ++t) {
    const Box tight = {TA[t * 4 + 0] - B2_POLYGON_RADIUS, TA[t * 4 + 1] - B2_POLYGON_RADIUS, TA[t * 4 + 2] + B2_POLYGON_RADIUS, TA[t * 4 + 3] + B2_POLYGON_RADIUS};
    w->tile_id[t] = w->tree.create_proxy(tight);
  }
  for (int c = 0; c < N; ++c) {
    // Car(world, angle, x, y): hull at the pose, wheels at UNROTATED offsets with the same angle (what k_install / the auto-reset do on the device)
    const double sa = H->spawn[c][0], sx = H->spawn[c][1], sy = H->spawn[c][2];
    const float fa = (float)sa;
    const Rot q = rot_of(fa);
    for (int k = 0; k < 5; ++k) {
      Xf x; x.q = q;
      x.p = v2(k == 0 ? (float)sx : (float)(sx + (k == 1 || k == 3 ? -55 : 55) * MCR_SIZE), k == 0 ? (float)sy : (float)(sy + (k <= 2 ? 80 : -82) * MCR_SIZE));
      w->xf[c * 5 + k] = x;
    }
    for (int f = 0; f < 8; ++f) w->fix_id[c * 8 + f] = w->tree.create_proxy(poly_box(fixture_poly(w->S, f), w->xf[c * 5 + fixture_body(f)]));
  }
  if (w->tree.node_count() > MCR_PID_LIMIT) return MCR_ERR_CAPACITY;
  H->pad0 = 1;                                                            // the blob carries proxy-id tables
  uint16_t* TP = (uint16_t*)(blob + MCR_OFF_TPID); uint16_t* FP = (uint16_t*)(blob + MCR_OFF_FPID);
  for (int t = 0; t < MCR_TILE_CAP; ++t) TP[t] = t < T ? (uint16_t)w->tile_id[t] : 0;
  for (int f = 0; f < MCR_MAX_AGENTS * 8; ++f) FP[f] = f < F ? (uint16_t)w->fix_id[f] : 0;
  return MCR_OK;
}

// The end of a world step: b2World::Solve -> b2Body::SynchronizeFixtures for every body, from the head of the world's body list (the body
// created LAST first: car N-1's wheels 3..0, its hull, car N-2's ..) and each body's fixture list from its head (hull: polygons 3..0) ->
// b2Fixture::Synchronize (the union of the fixture's boxes at the transforms the step was entered with and ends with) ->
// b2DynamicTree::MoveProxy (nothing while the fat box still holds it; else re-inserted fattened and stretched along the displacement).
// bodies: [N][5][6] f32 as mcr_get_state hands them out (c.x, c.y, angle, ..) after the step.
extern "C" int mcr_world_step(mcr_world* w, const float* bodies) {
  if (!w || !bodies) return MCR_ERR_ARG;
  if (w->fix_id.empty()) return MCR_ERR_STATE;
  const int N = w->N;
  std::vector<Xf> now((size_t)N * 5);
  for (int c = 0; c < N; ++c) for (int k = 0; k < 5; ++k) {
    const float* b = bodies + ((size_t)c * 5 + k) * 6;
    now[c * 5 + k] = xf_of(v2(b[0], b[1]), b[2], k == 0 ? v2(w->S.hull_lcx, w->S.hull_lcy) : v2(0.0f, 0.0f));
  }
  for (int c = N - 1; c >= 0; --c) for (int f = 7; f >= 0; --f) {
    const int k = fixture_body(f);
    const Xf& x1 = w->xf[c * 5 + k]; const Xf& x2 = now[c * 5 + k];
    const Box swept = box_union(poly_box(fixture_poly(w->S, f), x1), poly_box(fixture_poly(w->S, f), x2));
    const int id = w->fix_id[c * 8 + f];
    if (box_contains(w->tree.fat_box(id), swept)) continue;
    Box b = box_fatten(swept);
    const float dx = 2.0f * (x2.p.x - x1.p.x), dy = 2.0f * (x2.p.y - x1.p.y);          // b2_aabbMultiplier * displacement
    if (dx < 0.0f) b.lx += dx; else b.hx += dx;
    if (dy < 0.0f) b.ly += dy; else b.hy += dy;
    w->tree.move_proxy_to(id, b);
  }
  w->xf = now;
  return MCR_OK;
}

// the proxy ids of the live episode: out[0 .. T) tiles, then N * 8 car fixtures; returns the count written (tests)
extern "C" int mcr_world_proxy_ids(const mcr_world* w, int32_t* out, int cap) {
  if (!w || !out) return MCR_ERR_ARG;
  int n = 0;
  for (int id : w->tile_id) if (n < cap) out[n++] = id;
  for (int id : w->fix_id) if (n < cap) out[n++] = id;
  return n;
}
