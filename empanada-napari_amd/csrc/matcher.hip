// Host half of the 3-D stitching path: slice-to-slice label matching and the instance tracker in C++
// (reference: empanada/inference/matcher.py:136-326, patterns.py:55-121, tracker.py:61-123).
//
// The reference walks a stack of slices twice (forward, then backward with assign_new = False) and keeps every
// slice as a Python dict of {label: {'box','starts','runs'}}; objects that end up with one label are merged
// (merge_attrs: box union, join_ranges of the runs) and become the next target.
//
// Data model here (round 2): a slice is a set of COMPONENTS -- what the GPU run extractor delivers: the 8-connected
// components of one class, or the objects handed over by emp_sm_push_slice_objects -- stored once, as CSR run lists with
// area and box; an OBJECT is a label plus a list of member components.  Matching never touches runs twice:
//   * the run intersections between the components of two neighbouring slices are computed ONCE (grid-screened boxes,
//     two-pointer sweep over the sorted runs) into a sparse pair table that the forward AND the backward pass read;
//   * an object's area is the sum of its members' areas, its overlap with another object the sum over member pairs
//     (components of one slice are disjoint pixel sets: they come from one label map), its box the union;
//   * a step regroups member lists; runs are merged (sorted, touching runs joined: array_utils.py:658-699) only when an
//     object is read out -- by the tracker or by emp_sm_slice_object_runs.
// A step is
//   emp_sm_step_begin  -> sparse IoU (float64) / IoA (float32) of target x match objects; single-pair components of the
//                         overlap graph are assigned here, the rest is the dense SOLVER BLOCK,
//   [caller]           -> scipy.optimize.linear_sum_assignment on that block (the reference solves the whole matrix,
//                         matcher.py:218; zero entries never pass the threshold :226-229),
//   emp_sm_step_apply  -> label propagation, IoA merge, new labels, regrouping in first-occurrence order.
// Exactness notes: IoU = inter / (a + b - inter) in float64, IoA = float32(inter / area_match) as numpy stores it;
// objects are visited in dict order (ascending component label after extraction, first-occurrence order after a step).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include <thread>

#include "common.h"


namespace {

struct Pair { int a; int64_t inter; };        // overlap of a component with component `a` of the neighbouring slice

// A vector of trivially copyable T with N inline slots (round 4).  An object is nearly always ONE component and overlaps one
// or two targets, yet every object of every step owned two heap vectors: at thousands of objects per slice the allocator was
// most of a matching step (17 k malloc / free pairs per 4096^2 step).  Heap storage only beyond N elements.
template <typename T, int N>
class SmallVec {
 public:
  SmallVec() = default;
  SmallVec(size_t n, const T& v) { assign(n, v); }
  SmallVec(const SmallVec& o) { copy_from(o); }
  SmallVec(SmallVec&& o) noexcept { move_from(o); }
  SmallVec& operator=(const SmallVec& o) { if (this != &o) { release(); copy_from(o); } return *this; }
  SmallVec& operator=(SmallVec&& o) noexcept { if (this != &o) { release(); move_from(o); } return *this; }
  ~SmallVec() { release(); }
  size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }
  T* begin() { return data(); }
  T* end() { return data() + n_; }
  const T* begin() const { return data(); }
  const T* end() const { return data() + n_; }
  T& operator[](size_t i) { return data()[i]; }
  const T& operator[](size_t i) const { return data()[i]; }
  void clear() { n_ = 0; }
  void reserve(size_t c) { if (c > cap_) grow(c); }
  void push_back(const T& v) { if (n_ == cap_) grow(cap_ * 2); data()[n_++] = v; }
  void assign(size_t n, const T& v) { n_ = 0; reserve(n); for (size_t i = 0; i < n; ++i) data()[i] = v; n_ = (uint32_t)n; }
  template <typename It>
  void append(It first, It last) {
    const size_t k = (size_t)(last - first);
    reserve(n_ + k);
    T* d = data() + n_;
    for (; first != last; ++first) *d++ = *first;
    n_ += (uint32_t)k;
  }
  // std::vector's spelling, at the end only (what the matcher uses)
  template <typename It>
  void insert(const T* pos, It first, It last) { (void)pos; append(first, last); }

 private:
  T* data() { return heap_ ? heap_ : inl_; }
  const T* data() const { return heap_ ? heap_ : inl_; }
  void grow(size_t c) {
    if (c < (size_t)cap_ * 2) c = (size_t)cap_ * 2;
    T* h = static_cast<T*>(::operator new(c * sizeof(T)));
    std::memcpy(h, data(), n_ * sizeof(T));
    if (heap_) ::operator delete(heap_);
    heap_ = h;
    cap_ = (uint32_t)c;
  }
  void release() { if (heap_) ::operator delete(heap_); heap_ = nullptr; n_ = 0; cap_ = N; }
  void copy_from(const SmallVec& o) {
    heap_ = nullptr; cap_ = N; n_ = 0;
    reserve(o.n_);
    std::memcpy(data(), o.data(), o.n_ * sizeof(T));
    n_ = o.n_;
  }
  void move_from(SmallVec& o) {
    if (o.heap_) { heap_ = o.heap_; cap_ = o.cap_; n_ = o.n_; o.heap_ = nullptr; o.cap_ = N; o.n_ = 0; }
    else { heap_ = nullptr; cap_ = N; n_ = o.n_; std::memcpy(inl_, o.inl_, o.n_ * sizeof(T)); o.n_ = 0; }
  }
  T* heap_ = nullptr;
  uint32_t n_ = 0, cap_ = N;
  T inl_[N];
};

struct FObj {
  int64_t label;
  int64_t box[4];
  SmallVec<int, 2> members;                    // component indices of this slice
};

struct FSlice {
  int K = 0;                                   // components 0..K-1
  std::vector<int64_t> area, box;              // (K), (4K)
  std::vector<int64_t> run_off, starts, lens;  // CSR by component, runs ascending by start
  std::vector<FObj> objs;                      // current objects, in dict order
  std::vector<int> raster_comp;                // slices from the run extractor: component of every run in raster order
  // overlaps with the PREVIOUS pushed slice: by own component (fwd) and by the previous slice's component (rev)
  bool pairs_ready = false;
  std::vector<int64_t> fwd_off, rev_off;
  std::vector<Pair> fwd, rev;
};

struct Track {
  int64_t label;
  int64_t box[6];
  std::vector<int64_t> starts, runs;
  std::vector<int64_t> yz;   // yz stacks before finish(): (start in the (D,H) plane, length, x) triples
};

// ---- linear sum assignment -------------------------------------------------------------------------------------------
// The reference solves every slice's IoU matrix with scipy.optimize.linear_sum_assignment(maximize=True)
// (matcher.py:218).  WHICH optimal assignment comes out of tied matrices is a property of that implementation, and the
// label maps must be bit-identical, so this is scipy's algorithm restated step for step: the shortest-augmenting-path
// method of D. F. Crouse, "On implementing 2D rectangular assignment algorithms", IEEE Trans. Aerospace and Electronic
// Systems 52(4), 2016, as scipy >= 1.6 implements it (scipy/optimize/rectangular_lsap/rectangular_lsap.cpp; the build
// pins scipy 1.15.3) -- tall matrices are transposed, maximisation negates the costs, the column candidates are visited
// in DESCENDING index order (so that a constant matrix yields the identity), and among columns of equal reduced cost an
// unassigned one wins.  tests/test_lsa.py pins it against scipy itself on random, integer-valued (heavily tied), sparse
// and rectangular matrices.
int lsa_maximize(int64_t nr, int64_t nc, const double* cost_in, std::vector<int64_t>& rows, std::vector<int64_t>& cols) {
  rows.clear();
  cols.clear();
  if (nr == 0 || nc == 0) return 0;
  const bool transpose = nc < nr;
  std::vector<double> cost((size_t)(nr * nc));
  if (transpose) {
    for (int64_t i = 0; i < nr; ++i)
      for (int64_t j = 0; j < nc; ++j) cost[(size_t)(j * nr + i)] = -cost_in[i * nc + j];
    std::swap(nr, nc);
  } else {
    for (int64_t k = 0; k < nr * nc; ++k) cost[(size_t)k] = -cost_in[k];
  }
  for (double v : cost)
    if (v != v || v == -INFINITY) return -1;
  std::vector<double> u((size_t)nr, 0.0), v((size_t)nc, 0.0), spc((size_t)nc);
  std::vector<int64_t> path((size_t)nc, -1), col4row((size_t)nr, -1), row4col((size_t)nc, -1), remaining((size_t)nc);
  std::vector<char> SR((size_t)nr), SC((size_t)nc);
  for (int64_t cur = 0; cur < nr; ++cur) {
    // shortest augmenting path from row `cur`
    double min_val = 0.0;
    int64_t num_remaining = nc;
    for (int64_t it = 0; it < nc; ++it) remaining[(size_t)it] = nc - it - 1;
    std::fill(SR.begin(), SR.end(), 0);
    std::fill(SC.begin(), SC.end(), 0);
    std::fill(spc.begin(), spc.end(), INFINITY);
    int64_t sink = -1, i = cur;
    while (sink == -1) {
      int64_t index = -1;
      double lowest = INFINITY;
      SR[(size_t)i] = 1;
      for (int64_t it = 0; it < num_remaining; ++it) {
        const int64_t j = remaining[(size_t)it];
        const double r = min_val + cost[(size_t)(i * nc + j)] - u[(size_t)i] - v[(size_t)j];
        if (r < spc[(size_t)j]) {
          path[(size_t)j] = i;
          spc[(size_t)j] = r;
        }
        if (spc[(size_t)j] < lowest || (spc[(size_t)j] == lowest && row4col[(size_t)j] == -1)) {
          lowest = spc[(size_t)j];
          index = it;
        }
      }
      min_val = lowest;
      if (min_val == INFINITY) return -1;      // infeasible
      const int64_t j = remaining[(size_t)index];
      if (row4col[(size_t)j] == -1) sink = j;
      else i = row4col[(size_t)j];
      SC[(size_t)j] = 1;
      remaining[(size_t)index] = remaining[(size_t)--num_remaining];
    }
    // dual variables
    u[(size_t)cur] += min_val;
    for (int64_t r = 0; r < nr; ++r)
      if (SR[(size_t)r] && r != cur) u[(size_t)r] += min_val - spc[(size_t)col4row[(size_t)r]];
    for (int64_t j = 0; j < nc; ++j)
      if (SC[(size_t)j]) v[(size_t)j] -= min_val - spc[(size_t)j];
    // augment
    int64_t j = sink;
    while (true) {
      const int64_t r = path[(size_t)j];
      row4col[(size_t)j] = r;
      std::swap(col4row[(size_t)r], j);
      if (r == cur) break;
    }
  }
  rows.resize((size_t)nr);
  cols.resize((size_t)nr);
  if (transpose) {       // (rows of the transposed problem are the caller's columns) sorted by the caller's row
    std::vector<int64_t> order((size_t)nr);
    for (int64_t k = 0; k < nr; ++k) order[(size_t)k] = k;
    std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return col4row[(size_t)a] < col4row[(size_t)b]; });
    for (int64_t k = 0; k < nr; ++k) { rows[(size_t)k] = col4row[(size_t)order[(size_t)k]]; cols[(size_t)k] = order[(size_t)k]; }
  } else {
    for (int64_t k = 0; k < nr; ++k) { rows[(size_t)k] = k; cols[(size_t)k] = col4row[(size_t)k]; }
  }
  return 0;
}

// ---- the same algorithm on a SPARSE matrix --------------------------------------------------------------------------
// An IoU matrix is almost all zeros (an object overlaps a handful of objects of the next slice), and the dense algorithm
// above re-scans every remaining column in every Dijkstra step: O(nr * nc) per row, 0.13 ms on a 1024^2 slice of the bench
// stack and 340 ms on a 4096^2 slice with 7 600 objects.  This is the SAME algorithm -- the same floating-point values,
// the same comparisons, the same tie-breaking, hence the same assignment as scipy on the dense matrix -- with the scan
// replaced by an aggregate over the columns that have no non-zero entry in any row visited by the current search
// ("plain" columns).  For a plain column j every step s of the search offers the candidate fl(c_s - v_j) with
// c_s = fl(min_val_s - u_{i_s}) (its cost is -0.0), so spc_j = fl(C - v_j) with C = min_s c_s (fl is monotone): the
// minimum over the plain columns is attained at the largest v_j, and the scan-order rule of the dense loop ("a strictly
// lower value wins; among equal values the LAST unassigned column in scan order, else the FIRST column") is answered by a
// segment tree over the scan positions that keeps max v_j (all / unassigned columns): equality is tested on the rounded
// values fl(C - v), exactly as the dense loop compares them.  The few non-plain columns are kept in a list with explicit
// spc / path values.  `remaining` (the scan order) is the dense loop's array: reset to descending order per search,
// swap-removal per step -- mirrored by moving tree leaves, undone from a log when the search ends.
// tests/test_lsa.py pins it against scipy and against lsa_maximize on sparse, tied, rectangular matrices.
struct SparseLsa {
  int64_t nr = 0, nc = 0;            // rows = the smaller side (transposed if needed)
  bool transposed = false;
  std::vector<int64_t> adj_off;      // CSR by row: (column, cost = -w)
  std::vector<int64_t> adj_col;
  std::vector<double> adj_cost;
  // segment tree over scan positions (leaf p: the column at remaining[p]); absent leaves hold -inf
  int64_t leaves = 1;
  std::vector<double> T;             // node k: T[2k] = max v over present leaves, T[2k+1] = over present unassigned leaves (one cache line per node pair)
  std::vector<int64_t> remaining, pos_of;
  std::vector<double> u, v;
  std::vector<int64_t> col4row, row4col;

  void set_leaf(int64_t p, double val, bool unassigned) {
    int64_t k = leaves + p;
    T[2 * (size_t)k] = val;
    T[2 * (size_t)k + 1] = unassigned ? val : -INFINITY;
    // upwards until a node's two maxima come out unchanged: its ancestors are functions of it alone on this path (round 5:
    // most leaf updates of a search -- a column leaving the aggregate and coming back -- do not touch a subtree maximum;
    // the walk to the root was 2/3 of a matching step at thousands of objects per slice)
    for (k >>= 1; k >= 1; k >>= 1) {
      const double a = std::max(T[2 * (size_t)(2 * k)], T[2 * (size_t)(2 * k + 1)]);
      const double b = std::max(T[2 * (size_t)(2 * k) + 1], T[2 * (size_t)(2 * k + 1) + 1]);
      if (a == T[2 * (size_t)k] && b == T[2 * (size_t)k + 1]) break;
      T[2 * (size_t)k] = a;
      T[2 * (size_t)k + 1] = b;
    }
  }
  // first position whose value satisfies fl(C - val) <= t (pred is monotone in val: test the node maxima)
  template <typename P> int64_t first_pos(P pred) const {
    if (!pred(T[2])) return -1;
    int64_t k = 1;
    while (k < leaves) k = pred(T[2 * (size_t)(2 * k)]) ? 2 * k : 2 * k + 1;
    return k - leaves;
  }
  template <typename P> int64_t last_unassigned_pos(P pred) const {
    if (!pred(T[3])) return -1;
    int64_t k = 1;
    while (k < leaves) k = pred(T[2 * (size_t)(2 * k + 1) + 1]) ? 2 * k + 1 : 2 * k;
    return k - leaves;
  }
};

int lsa_maximize_sparse(int64_t R, int64_t Cn, int64_t nnz, const int64_t* er, const int64_t* ec, const double* ew,
                        std::vector<int64_t>& rows, std::vector<int64_t>& cols) {
  rows.clear();
  cols.clear();
  if (R == 0 || Cn == 0) return 0;
  SparseLsa S;
  S.transposed = Cn < R;
  S.nr = S.transposed ? Cn : R;
  S.nc = S.transposed ? R : Cn;
  const int64_t nr = S.nr, nc = S.nc;
  S.adj_off.assign((size_t)nr + 1, 0);
  for (int64_t k = 0; k < nnz; ++k) {
    if (!(ew[k] == ew[k]) || ew[k] == INFINITY || ew[k] == -INFINITY) return -1;
    const int64_t r = S.transposed ? ec[k] : er[k];
    if (r < 0 || r >= nr) return -1;
    ++S.adj_off[(size_t)r + 1];
  }
  for (int64_t r = 0; r < nr; ++r) S.adj_off[(size_t)r + 1] += S.adj_off[(size_t)r];
  S.adj_col.resize((size_t)nnz);
  S.adj_cost.resize((size_t)nnz);
  {
    std::vector<int64_t> fill(S.adj_off.begin(), S.adj_off.end() - 1);
    for (int64_t k = 0; k < nnz; ++k) {
      const int64_t r = S.transposed ? ec[k] : er[k], c = S.transposed ? er[k] : ec[k];
      if (c < 0 || c >= nc) return -1;
      const int64_t at = fill[(size_t)r]++;
      S.adj_col[(size_t)at] = c;
      S.adj_cost[(size_t)at] = -ew[k];
    }
  }
  while (S.leaves < nc) S.leaves <<= 1;
  S.T.assign((size_t)(4 * S.leaves), -INFINITY);
  S.remaining.resize((size_t)nc);
  S.pos_of.resize((size_t)nc);
  S.u.assign((size_t)nr, 0.0);
  S.v.assign((size_t)nc, 0.0);
  S.col4row.assign((size_t)nr, -1);
  S.row4col.assign((size_t)nc, -1);
  // base state of every search: position p holds column nc-1-p, every column plain and present
  for (int64_t p = 0; p < nc; ++p) {
    S.remaining[(size_t)p] = nc - 1 - p;
    S.pos_of[(size_t)(nc - 1 - p)] = p;
    S.T[2 * (size_t)(S.leaves + p)] = 0.0;
    S.T[2 * (size_t)(S.leaves + p) + 1] = 0.0;
  }
  for (int64_t k = S.leaves - 1; k >= 1; --k) {
    S.T[2 * (size_t)k] = std::max(S.T[2 * (size_t)(2 * k)], S.T[2 * (size_t)(2 * k + 1)]);
    S.T[2 * (size_t)k + 1] = std::max(S.T[2 * (size_t)(2 * k) + 1], S.T[2 * (size_t)(2 * k + 1) + 1]);
  }
  // per-search scratch (sized once)
  std::vector<char> nonplain((size_t)nc, 0), visited_col((size_t)nc, 0), touched((size_t)nc, 0);
  std::vector<double> spcx((size_t)nc, 0.0);          // explicit spc of non-plain columns; frozen spc of visited columns
  std::vector<int64_t> pathx((size_t)nc, -1);
  std::vector<int64_t> np_list, vis_cols, vis_rows, moved;     // moved: positions whose leaf / remaining entry changed
  std::vector<double> step_c;                                 // c_s of the steps of this search
  std::vector<int64_t> step_row;
  for (int64_t cur = 0; cur < nr; ++cur) {
    double min_val = 0.0, Cst = INFINITY;
    int64_t num_remaining = nc;
    np_list.clear(); vis_cols.clear(); vis_rows.clear(); moved.clear(); step_c.clear(); step_row.clear();
    int64_t sink = -1, i = cur;
    while (sink == -1) {
      vis_rows.push_back(i);
      const double ui = S.u[(size_t)i];
      const double cs = min_val - ui;                 // fl(fl(min_val + -0.0) - u_i)
      // the row's non-zero entries: their columns become non-plain (explicit spc from now on)
      for (int64_t a = S.adj_off[(size_t)i]; a < S.adj_off[(size_t)i + 1]; ++a) {
        const int64_t j = S.adj_col[(size_t)a];
        if (visited_col[(size_t)j]) continue;
        if (!nonplain[(size_t)j]) {
          nonplain[(size_t)j] = 1;
          np_list.push_back(j);
          // what the earlier steps of this search offered it as a plain column
          if (step_c.empty()) { spcx[(size_t)j] = INFINITY; pathx[(size_t)j] = -1; }
          else {
            const double val = Cst - S.v[(size_t)j];
            spcx[(size_t)j] = val;
            for (size_t q = 0; q < step_c.size(); ++q)
              if (step_c[q] - S.v[(size_t)j] == val) { pathx[(size_t)j] = step_row[q]; break; }
          }
          const int64_t p = S.pos_of[(size_t)j];
          S.set_leaf(p, -INFINITY, false);             // leaves the aggregate
          moved.push_back(p);
        }
        const double r = min_val + S.adj_cost[(size_t)a] - ui - S.v[(size_t)j];
        if (r < spcx[(size_t)j]) { spcx[(size_t)j] = r; pathx[(size_t)j] = i; }
        touched[(size_t)j] = 1;
      }
      // the zero entries of this row in the other non-plain columns
      for (int64_t j : np_list) {
        if (visited_col[(size_t)j]) continue;
        if (touched[(size_t)j]) { touched[(size_t)j] = 0; continue; }
        const double r = cs - S.v[(size_t)j];
        if (r < spcx[(size_t)j]) { spcx[(size_t)j] = r; pathx[(size_t)j] = i; }
      }
      step_c.push_back(cs);
      step_row.push_back(i);
      if (cs < Cst) Cst = cs;
      // the scan: minimum over the remaining columns with the dense loop's order rule ("a strictly lower value wins; among
      // equal values the LAST unassigned column in scan order, else the FIRST column").  The non-plain columns first: when
      // one of them is strictly below what the plain columns offer -- the usual case: an object with a real overlap --
      // the two tree descents are never looked at (round 5; they were half of a search)
      double npl = INFINITY;
      int64_t np_first = -1, np_last_un = -1;
      for (int64_t j : np_list) {
        if (visited_col[(size_t)j]) continue;
        const double val = spcx[(size_t)j];
        const int64_t p = S.pos_of[(size_t)j];
        const bool un = S.row4col[(size_t)j] == -1;
        if (np_first < 0 || val < npl) { npl = val; np_first = p; np_last_un = un ? p : -1; }
        else if (val == npl) {
          if (p < np_first) np_first = p;
          if (un && p > np_last_un) np_last_un = p;
        }
      }
      double lowest = npl;
      int64_t first_p = np_first, last_un_p = np_last_un;
      if (S.T[2] != -INFINITY) {                      // plain columns present
        const double t = Cst - S.T[2];
        if (np_first < 0 || t <= npl) {
          const double Cc = Cst;
          auto pred = [Cc, t](double val) { return val != -INFINITY && Cc - val <= t; };
          const int64_t tf = S.first_pos(pred), tl = S.last_unassigned_pos(pred);
          if (np_first < 0 || t < npl) { lowest = t; first_p = tf; last_un_p = tl; }
          else {      // equal: the candidates of both kinds compete by position
            if (tf >= 0 && tf < first_p) first_p = tf;
            if (tl > last_un_p) last_un_p = tl;
          }
        }
      }
      if (lowest == INFINITY) return -1;
      const int64_t index = last_un_p >= 0 ? last_un_p : first_p;
      const int64_t j = S.remaining[(size_t)index];
      if (!nonplain[(size_t)j]) {                      // a plain column: its path is the first step that offered this value
        for (size_t q = 0; q < step_c.size(); ++q)
          if (step_c[q] - S.v[(size_t)j] == lowest) { pathx[(size_t)j] = step_row[q]; break; }
      }
      spcx[(size_t)j] = lowest;                        // frozen: a visited column is never updated again
      min_val = lowest;
      visited_col[(size_t)j] = 1;
      vis_cols.push_back(j);
      if (S.row4col[(size_t)j] == -1) sink = j;
      else i = S.row4col[(size_t)j];
      // remaining[index] = remaining[--num_remaining]
      --num_remaining;
      const int64_t jl = S.remaining[(size_t)num_remaining];
      moved.push_back(index);
      moved.push_back(num_remaining);
      if (index != num_remaining) {
        S.remaining[(size_t)index] = jl;
        S.pos_of[(size_t)jl] = index;
        const bool present = !nonplain[(size_t)jl];      // (jl is not visited: it is still in `remaining`)
        S.set_leaf(index, present ? S.v[(size_t)jl] : -INFINITY, S.row4col[(size_t)jl] == -1);
      }
      S.set_leaf(num_remaining, -INFINITY, false);
    }
    // dual variables
    S.u[(size_t)cur] += min_val;
    for (int64_t r : vis_rows)
      if (r != cur) S.u[(size_t)r] += min_val - spcx[(size_t)S.col4row[(size_t)r]];
    for (int64_t j : vis_cols) S.v[(size_t)j] -= min_val - spcx[(size_t)j];
    // augment
    {
      int64_t j = sink;
      while (true) {
        const int64_t r = pathx[(size_t)j];
        S.row4col[(size_t)j] = r;
        std::swap(S.col4row[(size_t)r], j);
        if (r == cur) break;
      }
    }
    // back to the base state (descending scan order, every column plain) with the new duals / assignment flags
    for (int64_t j : np_list) nonplain[(size_t)j] = 0;
    for (int64_t j : vis_cols) visited_col[(size_t)j] = 0;
    for (int64_t p : moved) {
      const int64_t j = nc - 1 - p;
      S.remaining[(size_t)p] = j;
      S.pos_of[(size_t)j] = p;
    }
    for (int64_t p : moved) {
      const int64_t j = nc - 1 - p;
      S.set_leaf(p, S.v[(size_t)j], S.row4col[(size_t)j] == -1);
    }
    for (int64_t j : vis_cols) S.set_leaf(nc - 1 - j, S.v[(size_t)j], S.row4col[(size_t)j] == -1);
  }
  rows.resize((size_t)nr);
  cols.resize((size_t)nr);
  if (S.transposed) {
    std::vector<int64_t> order((size_t)nr);
    for (int64_t k = 0; k < nr; ++k) order[(size_t)k] = k;
    std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return S.col4row[(size_t)a] < S.col4row[(size_t)b]; });
    for (int64_t k = 0; k < nr; ++k) { rows[(size_t)k] = S.col4row[(size_t)order[(size_t)k]]; cols[(size_t)k] = order[(size_t)k]; }
  } else {
    for (int64_t k = 0; k < nr; ++k) { rows[(size_t)k] = k; cols[(size_t)k] = S.col4row[(size_t)k]; }
  }
  return 0;
}

int64_t intersection_sorted(const int64_t* s1, const int64_t* r1, size_t n1, const int64_t* s2, const int64_t* r2, size_t n2) {
  size_t i = 0, j = 0;
  int64_t acc = 0;
  while (i < n1 && j < n2) {
    const int64_t a0 = s1[i], a1 = a0 + r1[i], b0 = s2[j], b1 = b0 + r2[j];
    const int64_t lo = a0 > b0 ? a0 : b0, hi = a1 < b1 ? a1 : b1;
    if (hi > lo) acc += hi - lo;
    if (a1 < b1) ++i; else ++j;
  }
  return acc;
}

// component runs of a freshly built slice: sorted by start (extraction output is; pushed objects may not be)
void sort_component_runs(FSlice& sl) {
  for (int c = 0; c < sl.K; ++c) {
    const int64_t o0 = sl.run_off[(size_t)c], o1 = sl.run_off[(size_t)c + 1];
    bool sorted = true;
    for (int64_t i = o0 + 1; i < o1; ++i) if (sl.starts[(size_t)i] < sl.starts[(size_t)i - 1]) { sorted = false; break; }
    if (sorted) continue;
    std::vector<int64_t> idx((size_t)(o1 - o0));
    for (size_t i = 0; i < idx.size(); ++i) idx[i] = o0 + (int64_t)i;
    std::stable_sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b) { return sl.starts[(size_t)a] < sl.starts[(size_t)b]; });
    std::vector<int64_t> s(idx.size()), r(idx.size());
    for (size_t i = 0; i < idx.size(); ++i) { s[i] = sl.starts[(size_t)idx[i]]; r[i] = sl.lens[(size_t)idx[i]]; }
    std::copy(s.begin(), s.end(), sl.starts.begin() + o0);
    std::copy(r.begin(), r.end(), sl.lens.begin() + o0);
  }
}

// runs of an object: a single member's list as it is, several members' lists sorted and joined where they touch
void object_runs(const FSlice& sl, const FObj& o, std::vector<int64_t>& st, std::vector<int64_t>& rn) {
  st.clear();
  rn.clear();
  if (o.members.size() == 1) {
    const int c = o.members[0];
    st.assign(sl.starts.begin() + sl.run_off[(size_t)c], sl.starts.begin() + sl.run_off[(size_t)c + 1]);
    rn.assign(sl.lens.begin() + sl.run_off[(size_t)c], sl.lens.begin() + sl.run_off[(size_t)c + 1]);
    return;
  }
  // every member's list is ascending and the members are disjoint pixel sets (distinct starts): merge the lists pairwise
  // (two scratch buffers, ping-pong) instead of sorting their concatenation -- the tracker calls this for every
  // multi-member object of every slice
  static thread_local std::vector<std::pair<int64_t, int64_t>> bufa, bufb;
  bufa.clear();
  if (o.members.size() > 4) {      // many members (a label that collected dozens of fragments): one sort of the lot
    for (int c : o.members)
      for (int64_t i = sl.run_off[(size_t)c]; i < sl.run_off[(size_t)c + 1]; ++i)
        bufa.emplace_back(sl.starts[(size_t)i], sl.starts[(size_t)i] + sl.lens[(size_t)i]);
    std::sort(bufa.begin(), bufa.end());      // starts are distinct: no stability needed
  } else
  for (size_t m = 0; m < o.members.size(); ++m) {
    const int c = o.members[m];
    const size_t n0 = bufa.size();
    const int64_t o0 = sl.run_off[(size_t)c], o1 = sl.run_off[(size_t)c + 1];
    bool ordered = true;      // this member's runs all lie behind what is merged so far: append
    if (n0 > 0 && o1 > o0 && sl.starts[(size_t)o0] < bufa.back().first) ordered = false;
    if (ordered) {
      for (int64_t i = o0; i < o1; ++i) bufa.emplace_back(sl.starts[(size_t)i], sl.starts[(size_t)i] + sl.lens[(size_t)i]);
    } else {
      bufb.resize(n0 + (size_t)(o1 - o0));
      size_t a = 0, k = 0;
      int64_t b = o0;
      while (a < n0 && b < o1) {
        if (bufa[a].first <= sl.starts[(size_t)b]) bufb[k++] = bufa[a++];
        else { bufb[k++] = {sl.starts[(size_t)b], sl.starts[(size_t)b] + sl.lens[(size_t)b]}; ++b; }
      }
      while (a < n0) bufb[k++] = bufa[a++];
      for (; b < o1; ++b) bufb[k++] = {sl.starts[(size_t)b], sl.starts[(size_t)b] + sl.lens[(size_t)b]};
      bufa.swap(bufb);
    }
  }
  for (const auto& r : bufa) {
    if (!st.empty() && st.back() + rn.back() >= r.first) {
      const int64_t end = std::max(st.back() + rn.back(), r.second);
      rn.back() = end - st.back();
    } else {
      st.push_back(r.first);
      rn.push_back(r.second - r.first);
    }
  }
}

// sparse overlap table between the components of `prev` and `cur` (both directions)
void build_pairs(const FSlice& prev, FSlice& cur) {
  constexpr int CS = 6;                        // 64-pixel grid cells over the boxes
  int64_t my = 1, mx = 1;
  for (int c = 0; c < prev.K; ++c) { my = std::max(my, prev.box[4 * (size_t)c + 2]); mx = std::max(mx, prev.box[4 * (size_t)c + 3]); }
  for (int c = 0; c < cur.K; ++c) { my = std::max(my, cur.box[4 * (size_t)c + 2]); mx = std::max(mx, cur.box[4 * (size_t)c + 3]); }
  const int64_t gy = ((my - 1) >> CS) + 1, gx = ((mx - 1) >> CS) + 1;
  auto cells_of = [&](const int64_t* b, int64_t& y0, int64_t& y1, int64_t& x0, int64_t& x1) {
    y0 = std::max<int64_t>(0, b[0]) >> CS; x0 = std::max<int64_t>(0, b[1]) >> CS;
    y1 = std::min(std::max<int64_t>(b[0], b[2] - 1) >> CS, gy - 1);
    x1 = std::min(std::max<int64_t>(b[1], b[3] - 1) >> CS, gx - 1);
  };
  std::vector<int> cell_off((size_t)(gy * gx) + 1, 0);
  for (int c = 0; c < prev.K; ++c) {
    int64_t y0, y1, x0, x1;
    cells_of(&prev.box[4 * (size_t)c], y0, y1, x0, x1);
    for (int64_t y = y0; y <= y1; ++y) for (int64_t x = x0; x <= x1; ++x) ++cell_off[(size_t)(y * gx + x) + 1];
  }
  for (size_t i = 1; i < cell_off.size(); ++i) cell_off[i] += cell_off[i - 1];
  std::vector<int> items((size_t)cell_off.back());
  {
    std::vector<int> fill(cell_off.begin(), cell_off.end() - 1);
    for (int c = 0; c < prev.K; ++c) {
      int64_t y0, y1, x0, x1;
      cells_of(&prev.box[4 * (size_t)c], y0, y1, x0, x1);
      for (int64_t y = y0; y <= y1; ++y) for (int64_t x = x0; x <= x1; ++x) items[(size_t)fill[(size_t)(y * gx + x)]++] = c;
    }
  }
  cur.fwd_off.assign((size_t)cur.K + 1, 0);
  cur.fwd.clear();
  std::vector<int> seen((size_t)prev.K, -1);
  std::vector<int64_t> rev_cnt((size_t)prev.K + 1, 0);
  std::vector<Pair> tmp;
  for (int b = 0; b < cur.K; ++b) {
    const int64_t* bb = &cur.box[4 * (size_t)b];
    int64_t y0, y1, x0, x1;
    cells_of(bb, y0, y1, x0, x1);
    tmp.clear();
    const int64_t bo = cur.run_off[(size_t)b], bn = cur.run_off[(size_t)b + 1] - bo;
    for (int64_t y = y0; y <= y1; ++y)
      for (int64_t x = x0; x <= x1; ++x)
        for (int k = cell_off[(size_t)(y * gx + x)]; k < cell_off[(size_t)(y * gx + x) + 1]; ++k) {
          const int a = items[(size_t)k];
          if (seen[(size_t)a] == b) continue;
          seen[(size_t)a] = b;
          const int64_t* ab = &prev.box[4 * (size_t)a];
          // box screen (array_utils.py:148-211): non-empty intersection of the half-open boxes
          if (std::min(ab[2], bb[2]) <= std::max(ab[0], bb[0]) || std::min(ab[3], bb[3]) <= std::max(ab[1], bb[1])) continue;
          const int64_t ao = prev.run_off[(size_t)a], an = prev.run_off[(size_t)a + 1] - ao;
          const int64_t inter = intersection_sorted(&prev.starts[(size_t)ao], &prev.lens[(size_t)ao], (size_t)an,
                                                    &cur.starts[(size_t)bo], &cur.lens[(size_t)bo], (size_t)bn);
          if (inter > 0) tmp.push_back({a, inter});
        }
    std::sort(tmp.begin(), tmp.end(), [](const Pair& p, const Pair& q) { return p.a < q.a; });
    for (const Pair& pr : tmp) { cur.fwd.push_back(pr); ++rev_cnt[(size_t)pr.a + 1]; }
    cur.fwd_off[(size_t)b + 1] = (int64_t)cur.fwd.size();
  }
  cur.rev_off.assign((size_t)prev.K + 1, 0);
  for (int a = 0; a < prev.K; ++a) cur.rev_off[(size_t)a + 1] = cur.rev_off[(size_t)a] + rev_cnt[(size_t)a + 1];
  cur.rev.assign(cur.fwd.size(), Pair{0, 0});
  std::vector<int64_t> fill(cur.rev_off.begin(), cur.rev_off.end() - 1);
  for (int b = 0; b < cur.K; ++b)           // ascending b: every rev list comes out sorted by b
    for (int64_t k = cur.fwd_off[(size_t)b]; k < cur.fwd_off[(size_t)b + 1]; ++k)
      cur.rev[(size_t)fill[(size_t)cur.fwd[(size_t)k].a]++] = Pair{b, cur.fwd[(size_t)k].inter};
  cur.pairs_ready = true;
}

}  // namespace

struct emp_stack_matcher {
  int64_t class_id, divisor;
  double iou_thr, ioa_thr;
  bool do_match = true, assign_new = true;
  int64_t next_label = 0;
  int target_idx = -1;                         // slice whose objects are the current target (update_target)
  std::vector<FSlice> stack;
  // pending step: the overlap matrix is kept SPARSE (an object overlaps a handful of targets, not thousands):
  // col_ent[c] = (target index, intersection) of every target that intersects match object c, ascending target index
  int pending = -1, nt = 0, nm = 0;
  struct Ent { int t; int64_t inter; };
  std::vector<SmallVec<Ent, 2>> col_ent;
  std::vector<int64_t> ta, ma;                 // areas
  // the part of the assignment problem that needs a solver: rows / columns (ascending original indices) of the connected
  // components of the overlap graph that are not a single pair; `iou` is its dense block; the single pairs are applied
  // directly (an optimal assignment contains them: every other entry of their row and column is 0)
  std::vector<int> blk_rows, blk_cols;
  std::vector<int64_t> pair_rows, pair_cols;
  std::vector<double> iou;
  double iou_of(int r, int c) const {
    for (const Ent& e : col_ent[(size_t)c])
      if (e.t == r) return (double)e.inter / (double)(ta[(size_t)r] + ma[(size_t)c] - e.inter);
    return 0.0;
  }
  // step_apply's regrouping scratch: label -> group index (open addressing, cleared by epoch), group of every object
  std::vector<int64_t> ht_key, grp_label;
  std::vector<int> ht_val, grp_of, grp_count;
  std::vector<uint32_t> ht_stamp;
  uint32_t ht_epoch = 0;
  // scratch pair table for a target that is not the neighbouring slice (never on the pipeline's path)
  FSlice scratch;
  // tracker
  int axis = 0;
  int64_t D = 0, H = 0, W = 0;
  std::vector<Track> tracks;
  std::unordered_map<int64_t, size_t> track_of;
  bool finished = false;
  // assignment steps solved by lsa_maximize_sparse / handed to the dense solver (emp_sm_solver_stats)
  int64_t n_sparse_steps = 0, n_dense_steps = 0;
};

using namespace emp;

namespace {

void finish_slice(emp_stack_matcher* h, FSlice& sl, const std::vector<int64_t>& labels) {
  sort_component_runs(sl);
  sl.area.assign((size_t)sl.K, 0);
  for (int c = 0; c < sl.K; ++c) {
    int64_t a = 0;
    for (int64_t i = sl.run_off[(size_t)c]; i < sl.run_off[(size_t)c + 1]; ++i) a += sl.lens[(size_t)i];
    sl.area[(size_t)c] = a;
  }
  sl.objs.resize((size_t)sl.K);
  for (int c = 0; c < sl.K; ++c) {
    FObj& o = sl.objs[(size_t)c];
    o.label = labels[(size_t)c];
    std::memcpy(o.box, &sl.box[4 * (size_t)c], sizeof(o.box));
    o.members.assign(1, c);
  }
  (void)h;
}

int build_slice_from_runs(emp_stack_matcher* h, FSlice& sl, const int64_t* runs, int64_t n, int64_t width, int64_t id_offset) {
  // distinct labels, ascending: a counting pass when they are dense (component ids 1..K), a sort otherwise
  int64_t maxlab = 0;
  for (int64_t i = 0; i < n; ++i) {
    EMP_REQUIRE(runs[3 * i + 2] > 0, "sm_push_slice_runs: labels must be positive");
    maxlab = std::max(maxlab, runs[3 * i + 2]);
  }
  std::vector<int64_t> labels;
  std::vector<int> comp_of_run((size_t)n);
  if (maxlab <= 4 * n + 1024) {
    std::vector<int> idx((size_t)maxlab + 1, 0);
    for (int64_t i = 0; i < n; ++i) idx[(size_t)runs[3 * i + 2]] = 1;
    int k = 0;
    for (int64_t v = 1; v <= maxlab; ++v)
      if (idx[(size_t)v]) { idx[(size_t)v] = k++; labels.push_back(v + id_offset); }
    for (int64_t i = 0; i < n; ++i) comp_of_run[(size_t)i] = idx[(size_t)runs[3 * i + 2]];
  } else {
    std::vector<int64_t> u((size_t)n);
    for (int64_t i = 0; i < n; ++i) u[(size_t)i] = runs[3 * i + 2];
    std::sort(u.begin(), u.end());
    u.erase(std::unique(u.begin(), u.end()), u.end());
    for (int64_t v : u) labels.push_back(v + id_offset);
    for (int64_t i = 0; i < n; ++i)
      comp_of_run[(size_t)i] = (int)(std::lower_bound(u.begin(), u.end(), runs[3 * i + 2]) - u.begin());
  }
  const int K = (int)labels.size();
  sl.K = K;
  sl.run_off.assign((size_t)K + 1, 0);
  for (int64_t i = 0; i < n; ++i) ++sl.run_off[(size_t)comp_of_run[(size_t)i] + 1];
  for (int c = 0; c < K; ++c) sl.run_off[(size_t)c + 1] += sl.run_off[(size_t)c];
  sl.starts.resize((size_t)n);
  sl.lens.resize((size_t)n);
  sl.box.resize(4 * (size_t)K);
  for (int c = 0; c < K; ++c) {
    sl.box[4 * (size_t)c] = sl.box[4 * (size_t)c + 1] = INT64_MAX;
    sl.box[4 * (size_t)c + 2] = sl.box[4 * (size_t)c + 3] = INT64_MIN;
  }
  std::vector<int64_t> fill(sl.run_off.begin(), sl.run_off.end() - 1);
  for (int64_t i = 0; i < n; ++i) {          // raster order in, so every component's runs come out ascending
    const int c = comp_of_run[(size_t)i];
    const int64_t s = runs[3 * i], ln = runs[3 * i + 1];
    const int64_t e = s + ln - 1;
    const int64_t y0 = s / width, y1 = e / width;
    const int64_t x0 = y0 == y1 ? s % width : 0, x1 = y0 == y1 ? e % width : width - 1;
    int64_t* b = &sl.box[4 * (size_t)c];
    b[0] = std::min(b[0], y0); b[1] = std::min(b[1], x0);
    b[2] = std::max(b[2], y1 + 1); b[3] = std::max(b[3], x1 + 1);
    const int64_t pos = fill[(size_t)c]++;
    sl.starts[(size_t)pos] = s;
    sl.lens[(size_t)pos] = ln;
  }
  sl.raster_comp = std::move(comp_of_run);
  finish_slice(h, sl, labels);
  return EMP_OK;
}

// worker threads of the label-independent bulk work (slice construction, pair tables): EMP_SM_THREADS, default 4
int sm_threads() {
  static const int n = [] { const char* e = getenv("EMP_SM_THREADS"); const int v = e ? atoi(e) : 4; return v < 1 ? 1 : (v > 64 ? 64 : v); }();
  return n;
}

// fn(i) for i in [0, count) on up to sm_threads() threads (static interleaved split); returns the first non-zero result.
// The library's error text is thread-local (abi.hip): a worker's message (an EMP_REQUIRE inside fn) is captured on the
// worker and re-issued on the CALLING thread after the join, so that emp_last_error() names it; an exception thrown in a
// worker (std::bad_alloc) would otherwise reach std::terminate -- it is caught there and reported as EMP_ERR_NOMEM.
template <typename F>
int parallel_for(int64_t count, F fn) {
  const int T = (int)std::min<int64_t>(sm_threads(), count);
  if (T <= 1) {
    for (int64_t i = 0; i < count; ++i) { const int rc = fn(i); if (rc) return rc; }
    return EMP_OK;
  }
  std::vector<int> rcs((size_t)T, EMP_OK);
  std::vector<std::string> msgs((size_t)T);
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t] {
      try {
        for (int64_t i = t; i < count; i += T) {
          const int rc = fn(i);
          if (rc) { rcs[(size_t)t] = rc; msgs[(size_t)t] = emp_last_error(); return; }
        }
      } catch (const std::bad_alloc&) {
        rcs[(size_t)t] = EMP_ERR_NOMEM; msgs[(size_t)t] = "out of host memory in a matcher worker thread";
      } catch (const std::exception& e) {
        rcs[(size_t)t] = EMP_ERR_STATE; msgs[(size_t)t] = std::string("exception in a matcher worker thread: ") + e.what();
      }
    });
  for (auto& x : th) x.join();
  for (int t = 0; t < T; ++t)
    if (rcs[(size_t)t]) {
      set_error("%s", msgs[(size_t)t].c_str());
      return rcs[(size_t)t];
    }
  return EMP_OK;
}

}  // namespace

extern "C" {

emp_stack_matcher* emp_sm_create(int64_t class_id, int64_t label_divisor, double iou_thr, double ioa_thr, int do_match) {
  emp_stack_matcher* h = new emp_stack_matcher();
  h->class_id = class_id;
  h->divisor = label_divisor;
  h->iou_thr = iou_thr;
  h->ioa_thr = ioa_thr;
  h->do_match = do_match != 0;
  h->next_label = class_id * label_divisor + 1;
  return h;
}

void emp_sm_destroy(emp_stack_matcher* h) { delete h; }

// Append a slice from the run extractor: `runs` is (n,3) {start, length, label} in raster order, labels > 0; the slice
// plane is `width` pixels wide.  Components are created in ascending label order (regionprops order) with half-open
// boxes, exactly like rle.pan_seg_to_rle_seg; `id_offset` is added to every label (component index -> class id range).
int emp_sm_push_slice_runs(emp_stack_matcher* h, const int64_t* runs, int64_t n, int64_t width, int64_t id_offset) {
  EMP_REQUIRE(h && (runs || n == 0) && n >= 0 && width > 0, "sm_push_slice_runs: bad arguments");
  h->stack.emplace_back();
  return build_slice_from_runs(h, h->stack.back(), runs, n, width, id_offset);
}

// `count` slices at once (a launch group of the run extractor): runs[k] / n[k] as in emp_sm_push_slice_runs; the slices
// are built on worker threads (nothing in a slice depends on another)
int emp_sm_push_slices_runs(emp_stack_matcher* h, int64_t count, const int64_t* const* runs, const int64_t* n, int64_t width,
                            int64_t id_offset) {
  EMP_REQUIRE(h && count >= 0 && (count == 0 || (runs && n)) && width > 0, "sm_push_slices_runs: bad arguments");
  for (int64_t k = 0; k < count; ++k) EMP_REQUIRE(n[k] >= 0 && (runs[k] || n[k] == 0), "sm_push_slices_runs: bad slice %lld", (long long)k);
  const size_t base = h->stack.size();
  h->stack.resize(base + (size_t)count);
  const int rc = parallel_for(count, [&](int64_t k) { return build_slice_from_runs(h, h->stack[base + (size_t)k], runs[k], n[k], width, id_offset); });
  if (rc) h->stack.resize(base);
  return rc;
}

// Append a slice given as objects (labels, (n,4) boxes, CSR runs): the generic form of the above; every object is one
// component (objects of a slice must be disjoint pixel sets, as anything read from a label map is).
int emp_sm_push_slice_objects(emp_stack_matcher* h, int64_t n, const int64_t* labels, const int64_t* boxes,
                              const int64_t* off, const int64_t* starts, const int64_t* runs) {
  EMP_REQUIRE(h && n >= 0, "sm_push_slice_objects: bad arguments");
  h->stack.emplace_back();
  FSlice& sl = h->stack.back();
  sl.K = (int)n;
  sl.run_off.assign(off, off + n + 1);
  const int64_t base = n ? off[0] : 0;
  for (auto& v : sl.run_off) v -= base;
  const int64_t total = n ? off[n] - base : 0;
  sl.starts.assign(starts + base, starts + base + total);
  sl.lens.assign(runs + base, runs + base + total);
  sl.box.assign(boxes, boxes + 4 * n);
  std::vector<int64_t> labs(labels, labels + n);
  finish_slice(h, sl, labs);
  return EMP_OK;
}

int64_t emp_sm_num_slices(const emp_stack_matcher* h) { return h ? (int64_t)h->stack.size() : 0; }

// scipy.optimize.linear_sum_assignment(cost, maximize=True) on a dense row-major (nr, nc) float64 matrix: min(nr, nc)
// pairs, rows ascending -- the library's own solver (lsa_maximize above), exported so that tests can pin it against scipy
// the same on a sparse matrix: nnz entries (er[k], ec[k]) -> ew[k], every other entry 0; identical result to
// emp_lsa_maximize on the dense matrix (and so to scipy), min(nr, nc) pairs out
int emp_lsa_maximize_sparse(int64_t nr, int64_t nc, int64_t nnz, const int64_t* er, const int64_t* ec, const double* ew,
                            int64_t* rows, int64_t* cols) {
  EMP_REQUIRE(nr >= 0 && nc >= 0 && nnz >= 0 && (nnz == 0 || (er && ec && ew)) && ((rows && cols) || nr == 0 || nc == 0),
              "lsa_maximize_sparse: bad arguments");
  std::vector<int64_t> rr, cc;
  if (lsa_maximize_sparse(nr, nc, nnz, er, ec, ew, rr, cc) != 0) {
    set_error("lsa_maximize_sparse: invalid entries");
    return EMP_ERR_INVALID;
  }
  std::copy(rr.begin(), rr.end(), rows);
  std::copy(cc.begin(), cc.end(), cols);
  return EMP_OK;
}

int emp_lsa_maximize(const double* cost, int64_t nr, int64_t nc, int64_t* rows, int64_t* cols) {
  EMP_REQUIRE(nr >= 0 && nc >= 0 && (cost || nr * nc == 0) && ((rows && cols) || nr == 0 || nc == 0), "lsa_maximize: bad arguments");
  std::vector<int64_t> rr, cc;
  EMP_REQUIRE(lsa_maximize(nr, nc, cost, rr, cc) == 0, "lsa_maximize: cost matrix is infeasible (NaN or +inf entries)");
  std::copy(rr.begin(), rr.end(), rows);
  std::copy(cc.begin(), cc.end(), cols);
  return EMP_OK;
}

// ---- slab-wise matching (round 3: one matcher per rank, multigpu.py) --------------------------------------------------
// The passes of patterns.py:68-121 are a chain along the axis: slice z is matched against the RELABELLED slice z-1 (or
// z+1 on the way back).  What a slice hands to its neighbour is small -- the grouping of its components into labelled
// objects, in dict order, and the label counter -- while everything expensive (run intersections of neighbouring
// slices, run joins, the tracker) does not depend on labels.  A rank therefore keeps a matcher for its own slab plus one
// GHOST slice on either side (the neighbour's boundary slice, pushed from the same runs so that its components carry the
// same indices), builds every pair table up front, and only the state of a boundary slice travels down the ranks.

// Pair tables of slices (from, to]: the label-independent part of steps from+1 .. to, in any order, ahead of the chain.
int emp_sm_prepare(emp_stack_matcher* h, int64_t from, int64_t to) {
  EMP_REQUIRE(h && from >= 0 && to < (int64_t)h->stack.size(), "sm_prepare: bad range");
  if (!h->do_match) return EMP_OK;
  return parallel_for(to - from, [&](int64_t k) {      // slice i's tables are written by one thread, slice i-1 is only read
    const int64_t i = from + 1 + k;
    if (!h->stack[(size_t)i].pairs_ready) build_pairs(h->stack[(size_t)i - 1], h->stack[(size_t)i]);
    return (int)EMP_OK;
  });
}

int emp_sm_state_size(const emp_stack_matcher* h, int64_t idx, int64_t* n_obj, int64_t* n_mem) {
  EMP_REQUIRE(h && n_obj && n_mem && idx >= 0 && idx < (int64_t)h->stack.size(), "sm_state_size: bad arguments");
  const FSlice& sl = h->stack[(size_t)idx];
  *n_obj = (int64_t)sl.objs.size();
  int64_t m = 0;
  for (const FObj& o : sl.objs) m += (int64_t)o.members.size();
  *n_mem = m;
  return EMP_OK;
}

// The matching state of slice idx: its objects in dict order as (label, member components) -- CSR: off has n_obj + 1
// entries -- and the label counter (RLEMatcher.next_label).
int emp_sm_export_state(const emp_stack_matcher* h, int64_t idx, int64_t* labels, int64_t* off, int64_t* members,
                        int64_t* next_label) {
  EMP_REQUIRE(h && labels && off && members && next_label && idx >= 0 && idx < (int64_t)h->stack.size(),
              "sm_export_state: bad arguments");
  const FSlice& sl = h->stack[(size_t)idx];
  int64_t m = 0;
  for (size_t i = 0; i < sl.objs.size(); ++i) {
    labels[i] = sl.objs[i].label;
    off[i] = m;
    for (int c : sl.objs[i].members) members[m++] = c;
  }
  off[sl.objs.size()] = m;
  *next_label = h->next_label;
  return EMP_OK;
}

// Slice idx becomes the matcher's TARGET with the given objects (a ghost slice: the state its owner exported), the label
// counter is set (next_label < 0: kept) and assign_new chosen (1: forward pass, 0: backward pass, patterns.py:102-109).
int emp_sm_import_state(emp_stack_matcher* h, int64_t idx, int64_t n_obj, const int64_t* labels, const int64_t* off,
                        const int64_t* members, int64_t next_label, int assign_new) {
  EMP_REQUIRE(h && idx >= 0 && idx < (int64_t)h->stack.size() && n_obj >= 0 && (n_obj == 0 || (labels && off && members)),
              "sm_import_state: bad arguments");
  FSlice& sl = h->stack[(size_t)idx];
  std::vector<FObj> objs((size_t)n_obj);
  std::vector<char> used((size_t)sl.K, 0);
  for (int64_t i = 0; i < n_obj; ++i) {
    FObj& o = objs[(size_t)i];
    o.label = labels[i];
    EMP_REQUIRE(off[i + 1] > off[i], "sm_import_state: object %lld has no member", (long long)i);
    for (int64_t k = off[i]; k < off[i + 1]; ++k) {
      const int64_t c = members[k];
      EMP_REQUIRE(c >= 0 && c < sl.K && !used[(size_t)c], "sm_import_state: component %lld out of range or listed twice (the "
                  "ghost slice must be pushed from the runs its owner pushed)", (long long)c);
      used[(size_t)c] = 1;
      const int64_t* b = &sl.box[4 * (size_t)c];
      if (k == off[i]) std::memcpy(o.box, b, sizeof(o.box));
      else {
        o.box[0] = std::min(o.box[0], b[0]); o.box[1] = std::min(o.box[1], b[1]);
        o.box[2] = std::max(o.box[2], b[2]); o.box[3] = std::max(o.box[3], b[3]);
      }
      o.members.push_back((int)c);
    }
  }
  sl.objs.swap(objs);
  h->target_idx = (int)idx;
  h->pending = -1;
  if (next_label >= 0) h->next_label = next_label;
  h->assign_new = assign_new != 0;
  return EMP_OK;
}

// patterns.py:102-109: the backward pass starts from a fresh target and never creates labels
int emp_sm_begin_backward(emp_stack_matcher* h) {
  EMP_REQUIRE(h != nullptr, "sm_begin_backward: null handle");
  h->target_idx = -1;
  h->assign_new = false;
  return EMP_OK;
}

// First half of RLEMatcher.__call__ / apply_matchers for slice `idx` (see the header for the solver-block contract).
int emp_sm_step_begin(emp_stack_matcher* h, int64_t idx, int* nt, int* nm) {
  EMP_REQUIRE(h && nt && nm && idx >= 0 && idx < (int64_t)h->stack.size(), "sm_step_begin: bad arguments");
  *nt = -1;
  *nm = 0;
  h->pending = -1;
  if (!h->do_match) return EMP_OK;
  FSlice& cur = h->stack[(size_t)idx];
  if (h->target_idx < 0) {   // initialize_target (matcher.py:262-268)
    h->target_idx = (int)idx;
    if (!cur.objs.empty()) {
      int64_t mx = cur.objs[0].label;
      for (const FObj& o : cur.objs) mx = std::max(mx, o.label);
      h->next_label = mx + 1;
    }
    return EMP_OK;
  }
  EMP_REQUIRE(h->target_idx != (int)idx, "sm_step_begin: slice %lld is the current target", (long long)idx);
  FSlice& tgt = h->stack[(size_t)h->target_idx];
  h->pending = (int)idx;
  h->nt = (int)tgt.objs.size();
  h->nm = (int)cur.objs.size();
  *nt = h->nt;
  *nm = h->nm;
  h->blk_rows.clear(); h->blk_cols.clear(); h->pair_rows.clear(); h->pair_cols.clear(); h->iou.clear();
  h->col_ent.assign((size_t)h->nm, {});
  if (h->nt == 0 || h->nm == 0) return EMP_OK;
  const int nt_ = h->nt, nm_ = h->nm;
  // component-level overlaps between the two slices: stored with the LATER slice when they are neighbours (both passes
  // read the same table), computed into a scratch table otherwise
  const int ti = h->target_idx, ci = (int)idx;
  const FSlice* tab;        // table owner: pairs of (owner's previous slice, owner)
  bool target_is_prev;      // the target slice is the "previous" side of the table
  if (ci == ti + 1) {
    if (!cur.pairs_ready) build_pairs(tgt, cur);
    tab = &cur; target_is_prev = true;
  } else if (ci == ti - 1) {
    if (!tgt.pairs_ready) build_pairs(cur, tgt);
    tab = &tgt; target_is_prev = false;
  } else {
    h->scratch = FSlice();
    h->scratch.K = cur.K; h->scratch.box = cur.box; h->scratch.run_off = cur.run_off;
    h->scratch.starts = cur.starts; h->scratch.lens = cur.lens;
    build_pairs(tgt, h->scratch);
    tab = &h->scratch; target_is_prev = true;
  }
  h->ta.assign((size_t)nt_, 0);
  h->ma.assign((size_t)nm_, 0);
  std::vector<int> tobj((size_t)tgt.K, -1);
  for (int i = 0; i < nt_; ++i)
    for (int c : tgt.objs[(size_t)i].members) { tobj[(size_t)c] = i; h->ta[(size_t)i] += tgt.area[(size_t)c]; }
  std::vector<int> deg_r((size_t)nt_, 0);
  for (int j = 0; j < nm_; ++j) {
    auto& ents = h->col_ent[(size_t)j];
    for (int b : cur.objs[(size_t)j].members) {
      h->ma[(size_t)j] += cur.area[(size_t)b];
      const std::vector<int64_t>& off = target_is_prev ? tab->fwd_off : tab->rev_off;
      const std::vector<Pair>& lst = target_is_prev ? tab->fwd : tab->rev;
      for (int64_t k = off[(size_t)b]; k < off[(size_t)b + 1]; ++k) {
        const int t = tobj[(size_t)lst[(size_t)k].a];
        if (t < 0) continue;
        bool found = false;
        for (auto& e : ents) if (e.t == t) { e.inter += lst[(size_t)k].inter; found = true; break; }
        if (!found) ents.push_back({t, lst[(size_t)k].inter});
      }
    }
    std::sort(ents.begin(), ents.end(), [](const emp_stack_matcher::Ent& a, const emp_stack_matcher::Ent& b) { return a.t < b.t; });
    for (const auto& e : ents) ++deg_r[(size_t)e.t];
  }
  // does any object compete for a partner?  (degree > 1 on either side)
  bool any_conflict = false;
  for (int i = 0; i < nt_ && !any_conflict; ++i) any_conflict = deg_r[(size_t)i] > 1;
  for (int j = 0; j < nm_ && !any_conflict; ++j) any_conflict = h->col_ent[(size_t)j].size() > 1;
  // A slice with competing overlaps: the reference hands the WHOLE nt x nm matrix to scipy (matcher.py:216-218).  Round 2
  // restricted the solver to the conflict components (single pairs are in every optimal assignment): the same optimum
  // VALUE always and the same assignment whenever the optimum is unique, but with exactly tied IoU sums which optimum
  // scipy returns depends on the matrix it is given -- on tie-heavy synthetic matrices 17 % of the restricted solutions
  // differ from the full one (tests/test_lsa.py), and duplicate IoU values (1/2, 1/3 of few-pixel fragments) sit in nearly
  // every block of the bench stack.  So the whole matrix is solved -- by lsa_maximize_sparse, which is scipy's algorithm
  // with the same values, comparisons and tie-breaking on the matrix's non-zero entries (0.03 instead of 0.13 ms on a
  // 1024^2 slice with 390 objects, 3 instead of 340 ms on a 4096^2 slice with 7 600).
  // EMP_SM_FULL_LSA=1: the dense solve of the whole matrix (A/B, tests); =0: round 2's conflict block.
  const char* full_env = getenv("EMP_SM_FULL_LSA");
  int full_mode = !full_env ? 2 : (full_env[0] == '0' ? 0 : 1);      // 2 = sparse solve of the whole matrix
  {
    const char* e = getenv("EMP_SM_SCIPY");      // the caller solves with scipy itself: it gets the dense whole matrix, as the reference's call
    if (e && e[0] == '1' && full_mode == 2) full_mode = 1;
  }
  const bool full = full_mode == 1 && any_conflict;
  if (full_mode == 2 && any_conflict) {
    std::vector<int64_t> er, ec, rr, cc;
    std::vector<double> ew;
    for (int j = 0; j < nm_; ++j)
      for (const auto& e : h->col_ent[(size_t)j]) {
        er.push_back(e.t);
        ec.push_back(j);
        ew.push_back((double)e.inter / (double)(h->ta[(size_t)e.t] + h->ma[(size_t)j] - e.inter));
      }
    if (lsa_maximize_sparse(nt_, nm_, (int64_t)er.size(), er.data(), ec.data(), ew.data(), rr, cc) != 0) {
      set_error("sm_step_begin: the IoU matrix of slice %lld is not a valid cost matrix", (long long)idx);
      return EMP_ERR_INVALID;
    }
    // every pair is known: they travel as "single pairs" (step_apply thresholds them), nothing is left for a solver call
    h->pair_rows.assign(rr.begin(), rr.end());
    h->pair_cols.assign(cc.begin(), cc.end());
    ++h->n_sparse_steps;
    return EMP_OK;
  }
  if (full) {
    for (int i = 0; i < nt_; ++i) h->blk_rows.push_back(i);
    for (int j = 0; j < nm_; ++j) h->blk_cols.push_back(j);
  } else {
    // connected components of the overlap graph (only this mode -- round 2's conflict block, EMP_SM_FULL_LSA=0 -- and a
    // slice without any conflict, where every root is conflict-free, need them: the union-find ran in every step until
    // round 5): single pairs are assigned here, the rest forms the solver's block
    std::vector<int> parent((size_t)(nt_ + nm_));
    for (size_t i = 0; i < parent.size(); ++i) parent[i] = (int)i;
    auto find = [&](int a) { while (parent[(size_t)a] != a) { parent[(size_t)a] = parent[(size_t)parent[(size_t)a]]; a = parent[(size_t)a]; } return a; };
    std::vector<char> root_conflict((size_t)(nt_ + nm_), 0);
    if (any_conflict) {
      for (int j = 0; j < nm_; ++j)
        for (const auto& e : h->col_ent[(size_t)j]) {
          const int a = find(e.t), b2 = find(nt_ + j);
          if (a != b2) parent[(size_t)std::max(a, b2)] = std::min(a, b2);
        }
      for (int i = 0; i < nt_; ++i) if (deg_r[(size_t)i] > 1) root_conflict[(size_t)find(i)] = 1;
      for (int j = 0; j < nm_; ++j) if (h->col_ent[(size_t)j].size() > 1) root_conflict[(size_t)find(nt_ + j)] = 1;
    }
    for (int i = 0; i < nt_; ++i) if (deg_r[(size_t)i] > 0 && root_conflict[(size_t)find(i)]) h->blk_rows.push_back(i);
    for (int j = 0; j < nm_; ++j) {
      if (h->col_ent[(size_t)j].empty()) continue;
      if (root_conflict[(size_t)find(nt_ + j)]) h->blk_cols.push_back(j);
      else { h->pair_rows.push_back(h->col_ent[(size_t)j][0].t); h->pair_cols.push_back(j); }
    }
  }
  if (!h->blk_rows.empty()) {
    const size_t bt = h->blk_rows.size(), bm = h->blk_cols.size();
    std::vector<int> row_pos((size_t)nt_, -1);
    for (size_t i = 0; i < bt; ++i) row_pos[(size_t)h->blk_rows[i]] = (int)i;
    h->iou.assign(bt * bm, 0.0);
    for (size_t c = 0; c < bm; ++c) {
      const int j = h->blk_cols[c];
      for (const auto& e : h->col_ent[(size_t)j])
        h->iou[(size_t)row_pos[(size_t)e.t] * bm + c] = (double)e.inter / (double)(h->ta[(size_t)e.t] + h->ma[(size_t)j] - e.inter);
    }
  }
  return EMP_OK;
}

// dense float64 IoU matrix of the pending step's SOLVER BLOCK, emp_sm_pending_shape rows x columns (valid until the next
// step_begin): the rows / columns of the overlap graph's components that are not single pairs, in ascending order
const double* emp_sm_iou(const emp_stack_matcher* h) { return (h && !h->iou.empty()) ? h->iou.data() : nullptr; }

// Second half: `rows` / `cols` is the assignment on the solver block (all pairs; pairs below the IoU threshold are
// dropped here, matcher.py:226-229).
int emp_sm_step_apply(emp_stack_matcher* h, const int64_t* rows, const int64_t* cols, int64_t n) {
  EMP_REQUIRE(h && h->pending >= 0 && n >= 0, "sm_step_apply: no pending step");
  FSlice& cur = h->stack[(size_t)h->pending];
  const FSlice& tgt = h->stack[(size_t)h->target_idx];
  const int nt = h->nt, nm = h->nm;
  const bool have = nt > 0 && nm > 0;
  std::vector<int64_t> matched_t((size_t)nm, -1);     // per match object: index of the matched target, or -1
  if (have) {
    const int64_t bt = (int64_t)h->blk_rows.size(), bm = (int64_t)h->blk_cols.size();
    for (int64_t k = 0; k < n; ++k) {
      EMP_REQUIRE(rows[k] >= 0 && rows[k] < bt && cols[k] >= 0 && cols[k] < bm, "sm_step_apply: assignment out of range");
      const int r = h->blk_rows[(size_t)rows[k]], c = h->blk_cols[(size_t)cols[k]];
      if (h->iou[(size_t)rows[k] * (size_t)bm + (size_t)cols[k]] >= h->iou_thr) matched_t[(size_t)c] = r;
    }
    for (size_t k = 0; k < h->pair_rows.size(); ++k) {
      const int r = (int)h->pair_rows[k], c = (int)h->pair_cols[k];
      if (h->iou_of(r, c) >= h->iou_thr) matched_t[(size_t)c] = r;
    }
  }
  // new label per object, groups in first-occurrence order.  [Round 4: this loop and the regrouping below were the pole of
  // a step at thousands of objects per slice (1.0 of 1.6 ms at 7 600: one hash-map node, one member vector and several
  // re-allocations PER OBJECT); now two passes over flat arrays -- group index per object through an open-addressing
  // table kept in the handle, counts, then every group's member list sized once.]
  std::vector<int64_t>& group_label = h->grp_label;
  std::vector<int>& gidx = h->grp_of;
  group_label.clear();
  gidx.assign((size_t)nm, 0);
  size_t cap = 16;
  while (cap < (size_t)nm * 2 + 2) cap <<= 1;
  if (h->ht_key.size() < cap) { h->ht_key.assign(cap, 0); h->ht_val.assign(cap, 0); h->ht_stamp.assign(cap, 0); h->ht_epoch = 0; }
  if (++h->ht_epoch == 0) { std::fill(h->ht_stamp.begin(), h->ht_stamp.end(), 0u); h->ht_epoch = 1; }
  const size_t hmask = h->ht_key.size() - 1;
  const uint32_t epoch = h->ht_epoch;
  for (int c = 0; c < nm; ++c) {
    int64_t nl;
    if (matched_t[(size_t)c] >= 0) {
      nl = tgt.objs[(size_t)matched_t[(size_t)c]].label;
    } else {
      // argmax over the column of the float32 IoA matrix (first maximum; rows without overlap hold 0)
      float best = 0.f;
      int arg = 0;
      if (have)
        for (const auto& e : h->col_ent[(size_t)c]) {
          const float v = (float)((double)e.inter / (double)h->ma[(size_t)c]);
          if (v > best) { best = v; arg = e.t; }
        }
      // numpy >= 2 compares the float32 matrix entry with the Python float threshold in float32 (NEP 50)
      if (have && best >= (float)h->ioa_thr) nl = tgt.objs[(size_t)arg].label;
      else if (h->assign_new) nl = h->next_label++;
      else nl = cur.objs[(size_t)c].label;
    }
    size_t slot = ((uint64_t)nl * 0x9E3779B97F4A7C15ull >> 20) & hmask;
    while (h->ht_stamp[slot] == epoch && h->ht_key[slot] != nl) slot = (slot + 1) & hmask;
    if (h->ht_stamp[slot] != epoch) {
      h->ht_stamp[slot] = epoch;
      h->ht_key[slot] = nl;
      h->ht_val[slot] = (int)group_label.size();
      group_label.push_back(nl);
    }
    gidx[(size_t)c] = h->ht_val[slot];
  }
  const size_t G = group_label.size();
  std::vector<int>& gcount = h->grp_count;       // member COMPONENTS per group
  gcount.assign(G, 0);
  for (int c = 0; c < nm; ++c) gcount[(size_t)gidx[(size_t)c]] += (int)cur.objs[(size_t)c].members.size();
  std::vector<FObj> out(G);
  for (size_t g = 0; g < G; ++g) { out[g].label = group_label[g]; out[g].members.reserve((size_t)gcount[g]); }
  // merge_attrs folded over the group, objects in ascending order (= first-occurrence order inside a group): box union;
  // the runs are joined when the object is read out
  for (int c = 0; c < nm; ++c) {
    FObj& o = out[(size_t)gidx[(size_t)c]];
    const FObj& sobj = cur.objs[(size_t)c];
    if (o.members.empty()) {
      std::memcpy(o.box, sobj.box, sizeof(o.box));
    } else {
      o.box[0] = std::min(o.box[0], sobj.box[0]); o.box[1] = std::min(o.box[1], sobj.box[1]);
      o.box[2] = std::max(o.box[2], sobj.box[2]); o.box[3] = std::max(o.box[3], sobj.box[3]);
    }
    o.members.insert(o.members.end(), sobj.members.begin(), sobj.members.end());
  }
  cur.objs.swap(out);
  h->target_idx = h->pending;      // update_target
  h->pending = -1;
  return EMP_OK;
}

// Runs `count` consecutive steps from slice idx in direction dir (+1 forward pass, -1 backward pass; `track`: feed the
// tracker after every step, as backward_matching does) and stops BEFORE the assignment of the first slice with a
// non-empty solver block: *stopped_at = that slice (its step is pending: the caller fetches emp_sm_iou, solves, calls
// emp_sm_step_apply [+ emp_sm_track] and resumes behind it), or -1 when all steps ran.
int emp_sm_track(emp_stack_matcher* h, int64_t idx, int64_t index2d);
int emp_sm_run(emp_stack_matcher* h, int64_t idx, int dir, int64_t count, int track, int64_t* stopped_at) {
  EMP_REQUIRE(h && stopped_at && (dir == 1 || dir == -1) && count >= 0, "sm_run: bad arguments");
  *stopped_at = -1;
  for (int64_t k = 0; k < count; ++k) {
    const int64_t i = idx + k * dir;
    int nt = 0, nm = 0;
    int rc = emp_sm_step_begin(h, i, &nt, &nm);
    if (rc) return rc;
    if (nt >= 0) {
      if (!h->blk_rows.empty()) {
        // a real assignment problem.  Round 3: solved here (lsa_maximize = scipy's algorithm restated); with
        // EMP_SM_SCIPY=1 the step is handed back to the caller, who solves it with scipy itself as round 2 did
        const char* e = getenv("EMP_SM_SCIPY");      // read per call: tests run both solvers in one process
        if (e && e[0] == '1') { *stopped_at = i; return EMP_OK; }
        std::vector<int64_t> rr, cc;
        ++h->n_dense_steps;
        if (lsa_maximize((int64_t)h->blk_rows.size(), (int64_t)h->blk_cols.size(), h->iou.data(), rr, cc) != 0) {
          set_error("sm_run: the IoU block of slice %lld is not a valid cost matrix", (long long)i);
          return EMP_ERR_INVALID;
        }
        rc = emp_sm_step_apply(h, rr.data(), cc.data(), (int64_t)rr.size());
      } else {
        rc = emp_sm_step_apply(h, nullptr, nullptr, 0);
      }
      if (rc) return rc;
    }
    if (track) {
      rc = emp_sm_track(h, i, i);
      if (rc) return rc;
    }
  }
  return EMP_OK;
}

// how the steps with competing overlaps were solved so far: by the sparse solver inside step_begin / by a dense solver call
int emp_sm_solver_stats(const emp_stack_matcher* h, int64_t* sparse_steps, int64_t* dense_steps) {
  EMP_REQUIRE(h && sparse_steps && dense_steps, "sm_solver_stats: bad arguments");
  *sparse_steps = h->n_sparse_steps;
  *dense_steps = h->n_dense_steps;
  return EMP_OK;
}

int emp_sm_pending_shape(const emp_stack_matcher* h, int* nt, int* nm) {
  EMP_REQUIRE(h && nt && nm && h->pending >= 0, "sm_pending_shape: no pending step");
  *nt = (int)h->blk_rows.size();      // the solver block (0 x 0: nothing to solve, call emp_sm_step_apply with n = 0)
  *nm = (int)h->blk_cols.size();
  return EMP_OK;
}

// ---- tracker (tracker.py:61-123) -------------------------------------------------------------------------------
int emp_sm_tracker_init(emp_stack_matcher* h, int axis, int64_t D, int64_t H, int64_t W) {
  EMP_REQUIRE(h && axis >= 0 && axis <= 2 && D > 0 && H > 0 && W > 0, "sm_tracker_init: bad arguments");
  h->axis = axis; h->D = D; h->H = H; h->W = W;
  h->tracks.clear();
  h->track_of.clear();
  h->finished = false;
  return EMP_OK;
}

// InstanceTracker.update for a slice that came from the run extractor, in ONE pass over its runs in raster order: every
// run goes to the track of the object its component belongs to, joined to that object's previous run of THIS slice where
// they touch.  Same lists as object_runs() builds per object (a member's own runs never touch -- the extractor joined them
// -- and the members' lists merged by start ARE the raster order), without a sort or a merge per object.
static int track_raster(emp_stack_matcher* h, const FSlice& sl, int64_t index2d) {
  const int64_t H = h->H, W = h->W;
  static thread_local std::vector<int> obj_of_comp;
  static thread_local std::vector<int64_t> cursor, last_start, last_end;
  static thread_local std::vector<size_t> tidx;
  const size_t no = sl.objs.size();
  obj_of_comp.assign((size_t)sl.K, -1);
  cursor.assign(sl.run_off.begin(), sl.run_off.end() - 1);
  tidx.resize(no);
  last_start.assign(no, 0);
  last_end.assign(no, INT64_MIN);               // no run of this slice yet
  for (size_t k = 0; k < no; ++k) {
    const FObj& o = sl.objs[k];
    int64_t box[6];
    const int64_t y1 = o.box[0], x1 = o.box[1], y2 = o.box[2], x2 = o.box[3];
    if (h->axis == 0) { box[0] = index2d; box[1] = y1; box[2] = x1; box[3] = index2d + 1; box[4] = y2; box[5] = x2; }
    else if (h->axis == 1) { box[0] = y1; box[1] = index2d; box[2] = x1; box[3] = y2; box[4] = index2d + 1; box[5] = x2; }
    else { box[0] = y1; box[1] = x1; box[2] = index2d; box[3] = y2; box[4] = x2; box[5] = index2d + 1; }
    auto it = h->track_of.find(o.label);
    if (it == h->track_of.end()) {
      h->track_of.emplace(o.label, h->tracks.size());
      tidx[k] = h->tracks.size();
      h->tracks.emplace_back();
      Track& t = h->tracks.back();
      t.label = o.label;
      std::memcpy(t.box, box, sizeof(box));
    } else {
      tidx[k] = it->second;
      Track& t = h->tracks[it->second];
      for (int q = 0; q < 3; ++q) { t.box[q] = std::min(t.box[q], box[q]); t.box[3 + q] = std::max(t.box[3 + q], box[3 + q]); }
    }
    size_t nruns = 0;
    for (int c : o.members) {
      obj_of_comp[(size_t)c] = (int)k;
      nruns += (size_t)(sl.run_off[(size_t)c + 1] - sl.run_off[(size_t)c]);
    }
    Track& t = h->tracks[tidx[k]];
    if (h->axis == 2) {
      if (t.yz.capacity() < t.yz.size() + 3 * nruns) t.yz.reserve(std::max(t.yz.size() + 3 * nruns, 2 * t.yz.capacity()));      // (geometric: an exact reserve per slice re-copies the list every time)
    } else if (t.starts.capacity() < t.starts.size() + nruns) {
      const size_t want = std::max(t.starts.size() + nruns, 2 * t.starts.capacity());
      t.starts.reserve(want);
      t.runs.reserve(want);
    }
  }
  const int axis = h->axis;
  const int64_t plane = H * W;
  const size_t n = sl.raster_comp.size();
  for (size_t i = 0; i < n; ++i) {
    const int c = sl.raster_comp[i];
    const int k = obj_of_comp[(size_t)c];
    const int64_t pos = cursor[(size_t)c]++;
    if (k < 0) continue;
    const int64_t s = sl.starts[(size_t)pos], e = s + sl.lens[(size_t)pos];
    Track& t = h->tracks[tidx[(size_t)k]];
    if (last_end[(size_t)k] >= s) {               // touches (or overlaps) the object's previous run of this slice
      if (e > last_end[(size_t)k]) {
        last_end[(size_t)k] = e;
        if (axis == 2) t.yz[t.yz.size() - 2] = e - last_start[(size_t)k];
        else t.runs.back() = e - last_start[(size_t)k];
      }
      continue;
    }
    last_start[(size_t)k] = s;
    last_end[(size_t)k] = e;
    if (axis == 0) {
      t.starts.push_back(s + index2d * plane);
      t.runs.push_back(e - s);
    } else if (axis == 1) {
      t.starts.push_back((s / W) * plane + index2d * W + (s % W));
      t.runs.push_back(e - s);
    } else {
      t.yz.push_back(s); t.yz.push_back(e - s); t.yz.push_back(index2d);
    }
  }
  return EMP_OK;
}

// InstanceTracker.update for slice idx of the stack at position index2d along the axis
int emp_sm_track(emp_stack_matcher* h, int64_t idx, int64_t index2d) {
  EMP_REQUIRE(h && !h->finished && idx >= 0 && idx < (int64_t)h->stack.size(), "sm_track: bad arguments");
  const int64_t H = h->H, W = h->W;
  const FSlice& sl = h->stack[(size_t)idx];
  if (!sl.raster_comp.empty() && sl.raster_comp.size() == sl.starts.size()) return track_raster(h, sl, index2d);
  std::vector<int64_t> ost, orn;
  for (const FObj& o : sl.objs) {
    int64_t box[6];
    const int64_t y1 = o.box[0], x1 = o.box[1], y2 = o.box[2], x2 = o.box[3];
    if (h->axis == 0) { box[0] = index2d; box[1] = y1; box[2] = x1; box[3] = index2d + 1; box[4] = y2; box[5] = x2; }
    else if (h->axis == 1) { box[0] = y1; box[1] = index2d; box[2] = x1; box[3] = y2; box[4] = index2d + 1; box[5] = x2; }
    else { box[0] = y1; box[1] = x1; box[2] = index2d; box[3] = y2; box[4] = x2; box[5] = index2d + 1; }
    auto it = h->track_of.find(o.label);
    Track* t;
    if (it == h->track_of.end()) {
      h->track_of.emplace(o.label, h->tracks.size());
      h->tracks.emplace_back();
      t = &h->tracks.back();
      t->label = o.label;
      std::memcpy(t->box, box, sizeof(box));
    } else {
      t = &h->tracks[it->second];
      for (int k = 0; k < 3; ++k) { t->box[k] = std::min(t->box[k], box[k]); t->box[3 + k] = std::max(t->box[3 + k], box[3 + k]); }
    }
    // the object's runs: a single member's component list is used where it lies (the common case), several members'
    // lists are sorted and joined first (merge_attrs)
    const int64_t* ps;
    const int64_t* pr;
    size_t n;
    if (o.members.size() == 1) {
      const size_t c = (size_t)o.members[0];
      ps = sl.starts.data() + sl.run_off[c];
      pr = sl.lens.data() + sl.run_off[c];
      n = (size_t)(sl.run_off[c + 1] - sl.run_off[c]);
    } else {
      object_runs(sl, o, ost, orn);
      ps = ost.data(); pr = orn.data(); n = ost.size();
    }
    if (h->axis == 0) {                       // plane (H,W) at depth index2d
      const int64_t base = index2d * (H * W);
      const size_t at = t->starts.size();
      t->starts.resize(at + n);
      t->runs.insert(t->runs.end(), pr, pr + n);
      int64_t* dst = t->starts.data() + at;
      for (size_t i = 0; i < n; ++i) dst[i] = ps[i] + base;
    } else if (h->axis == 1) {                // plane (D,W) at row index2d: runs along x stay runs
      const size_t at = t->starts.size();
      t->starts.resize(at + n);
      t->runs.insert(t->runs.end(), pr, pr + n);
      int64_t* dst = t->starts.data() + at;
      for (size_t i = 0; i < n; ++i) dst[i] = (ps[i] / W) * (H * W) + index2d * W + (ps[i] % W);
    } else {                                  // plane (D,H) at column index2d: every voxel is its own run;
      const size_t at = t->yz.size();         // kept as the 2-D run until finish() turns the object around
      t->yz.resize(at + 3 * n);
      int64_t* dst = t->yz.data() + at;
      for (size_t i = 0; i < n; ++i) { dst[3 * i] = ps[i]; dst[3 * i + 1] = pr[i]; dst[3 * i + 2] = index2d; }
    }
  }
  return EMP_OK;
}

// InstanceTracker.update for local slices last, last-1, ..., first (the order backward_matching feeds the tracker,
// patterns.py:102-134) at positions global_first + (i - first).  Sizes every track once (an upper bound: joins only
// shrink a list) before the slices are walked, so no list is regrown on the way.
int emp_sm_track_range(emp_stack_matcher* h, int64_t first, int64_t last, int64_t global_first) {
  EMP_REQUIRE(h && !h->finished && first >= 0 && last < (int64_t)h->stack.size(), "sm_track_range: bad arguments");
  if (h->axis != 2) {
    std::unordered_map<int64_t, size_t> need;
    for (int64_t i = last; i >= first; --i) {
      const FSlice& sl = h->stack[(size_t)i];
      for (const FObj& o : sl.objs) {
        size_t n = 0;
        for (int c : o.members) n += (size_t)(sl.run_off[(size_t)c + 1] - sl.run_off[(size_t)c]);
        need[o.label] += n;
      }
    }
    for (int64_t i = last; i >= first; --i)
      for (const FObj& o : h->stack[(size_t)i].objs) {
        auto it = need.find(o.label);
        if (it == need.end() || it->second == 0) continue;
        auto tk = h->track_of.find(o.label);
        if (tk == h->track_of.end()) {       // create in first-seen order, exactly where emp_sm_track would
          h->track_of.emplace(o.label, h->tracks.size());
          h->tracks.emplace_back();
          Track& t = h->tracks.back();
          t.label = o.label;
          t.box[0] = t.box[1] = t.box[2] = INT64_MAX;
          t.box[3] = t.box[4] = t.box[5] = INT64_MIN;
          tk = h->track_of.find(o.label);
        }
        Track& t = h->tracks[tk->second];
        t.starts.reserve(t.starts.size() + it->second);
        t.runs.reserve(t.runs.size() + it->second);
        it->second = 0;
      }
  }
  for (int64_t i = last; i >= first; --i) {
    const int rc = emp_sm_track(h, i, global_first + (i - first));
    if (rc != EMP_OK) return rc;
  }
  return EMP_OK;
}

int emp_sm_tracker_finish(emp_stack_matcher* h) {
  EMP_REQUIRE(h != nullptr, "sm_tracker_finish: null handle");
  if (h->axis == 2) {
    // tracker.py:84-88,111-120: every voxel of a yz object becomes a run of 1 at its raveled (z,y,x) index, the
    // indices are sorted and re-encoded.  Same result without the sort: the object's voxels are marked in a bitmap
    // over its bounding box (rows = (z,y), bits = x) and the rows are scanned in order; a run that ends at the last
    // column continues into column 0 of the next row exactly when the raveled indices are consecutive.
    const int64_t H = h->H, W = h->W;
    std::vector<uint64_t> bits;
    for (Track& t : h->tracks) {
      const int64_t z0 = t.box[0], y0 = t.box[1], x0 = t.box[2];
      const int64_t nz = t.box[3] - z0, ny = t.box[4] - y0, nx = t.box[5] - x0;
      const int64_t wpr = (nx + 63) / 64;                    // words per row
      const int64_t words = nz * ny * wpr;
      t.starts.clear();
      t.runs.clear();
      if (words > ((int64_t)1 << 27)) {                      // > 1 GiB of bitmap: sort the voxels instead
        std::vector<int64_t> v;
        for (size_t i = 0; i < t.yz.size(); i += 3)
          for (int64_t f = t.yz[i]; f < t.yz[i] + t.yz[i + 1]; ++f) v.push_back((f / H) * (H * W) + (f % H) * W + t.yz[i + 2]);
        std::sort(v.begin(), v.end());
        for (size_t i = 0; i < v.size(); ++i) {
          if (i > 0 && v[i] == v[i - 1] + 1) ++t.runs.back();
          else { t.starts.push_back(v[i]); t.runs.push_back(1); }
        }
      } else {
        bits.assign((size_t)words, 0);
        for (size_t i = 0; i < t.yz.size(); i += 3) {
          const int64_t f0 = t.yz[i], len = t.yz[i + 1], bx = t.yz[i + 2] - x0;
          int64_t z = f0 / H, y = f0 % H;
          int64_t row = (z - z0) * ny + (y - y0);
          const int64_t wofs = bx >> 6;
          const uint64_t bit = (uint64_t)1 << (bx & 63);
          for (int64_t k = 0; k < len; ++k) {
            bits[(size_t)(row * wpr + wofs)] |= bit;
            if (++y == H) { y = 0; ++z; row = (z - z0) * ny - y0; }   // a 2-D run may wrap into the next z row
            else ++row;
          }
        }
        const uint64_t* wp = bits.data();
        for (int64_t z = 0; z < nz; ++z)
          for (int64_t y = 0; y < ny; ++y, wp += wpr) {
            const int64_t row0 = (z + z0) * (H * W) + (y + y0) * W + x0;
            int64_t open = -1;                               // first column of the run being scanned
            for (int64_t wi = 0; wi < wpr; ++wi) {
              uint64_t m = wp[wi];
              const int64_t c0 = wi * 64;
              if (m == 0) {
                if (open >= 0) { t.starts.push_back(row0 + open); t.runs.push_back(c0 - open); open = -1; }
                continue;
              }
              if (m == ~(uint64_t)0) { if (open < 0) open = c0; continue; }
              int64_t pos = 0;
              while (pos < 64) {
                if (open < 0) {                               // look for the next set bit
                  const uint64_t rest = m >> pos;
                  if (rest == 0) break;
                  pos += __builtin_ctzll(rest);
                  open = c0 + pos;
                } else {                                      // look for the next clear bit
                  const uint64_t inv = (~m) >> pos;
                  if (inv == 0) break;                        // the run continues into the next word
                  pos += __builtin_ctzll(inv);
                  t.starts.push_back(row0 + open);
                  t.runs.push_back(c0 + pos - open);
                  open = -1;
                }
              }
            }
            if (open >= 0) { t.starts.push_back(row0 + open); t.runs.push_back(wpr * 64 - open); }
          }
        // (padding bits of a row are never set, so a run is still open at the end of a row only when nx % 64 == 0)
        // runs that touch in raveled order (column W-1 -> column 0 of the next row) are joined here
        size_t o = 0;
        for (size_t i = 0; i < t.starts.size(); ++i) {
          if (o > 0 && t.starts[o - 1] + t.runs[o - 1] == t.starts[i]) t.runs[o - 1] += t.runs[i];
          else { t.starts[o] = t.starts[i]; t.runs[o] = t.runs[i]; ++o; }
        }
        t.starts.resize(o);
        t.runs.resize(o);
      }
      std::vector<int64_t>().swap(t.yz);
    }
  }
  h->finished = true;
  return EMP_OK;
}

int64_t emp_sm_num_tracks(const emp_stack_matcher* h) { return h ? (int64_t)h->tracks.size() : 0; }

int emp_sm_track_info(const emp_stack_matcher* h, int64_t k, int64_t* label, int64_t* box6, int64_t* n_runs) {
  EMP_REQUIRE(h && k >= 0 && k < (int64_t)h->tracks.size() && label && box6 && n_runs, "sm_track_info: bad arguments");
  const Track& t = h->tracks[(size_t)k];
  *label = t.label;
  std::memcpy(box6, t.box, sizeof(t.box));
  *n_runs = (int64_t)t.starts.size();
  return EMP_OK;
}

int emp_sm_track_runs(const emp_stack_matcher* h, int64_t k, int64_t* starts, int64_t* runs) {
  EMP_REQUIRE(h && k >= 0 && k < (int64_t)h->tracks.size() && starts && runs, "sm_track_runs: bad arguments");
  const Track& t = h->tracks[(size_t)k];
  std::memcpy(starts, t.starts.data(), t.starts.size() * sizeof(int64_t));
  std::memcpy(runs, t.runs.data(), t.runs.size() * sizeof(int64_t));
  return EMP_OK;
}

// All tracks at once, in first-seen order: labels (T), boxes (T,6), run counts (T); then the run lists back to back.
int emp_sm_tracks_info(const emp_stack_matcher* h, int64_t* labels, int64_t* boxes6, int64_t* counts, int64_t* total_runs) {
  EMP_REQUIRE(h && total_runs, "sm_tracks_info: bad arguments");
  int64_t tot = 0;
  for (size_t k = 0; k < h->tracks.size(); ++k) {
    const Track& t = h->tracks[k];
    if (labels) labels[k] = t.label;
    if (boxes6) std::memcpy(boxes6 + 6 * k, t.box, sizeof(t.box));
    if (counts) counts[k] = (int64_t)t.starts.size();
    tot += (int64_t)t.starts.size();
  }
  *total_runs = tot;
  return EMP_OK;
}

int emp_sm_tracks_runs(const emp_stack_matcher* h, int64_t* starts, int64_t* runs) {
  EMP_REQUIRE(h && starts && runs, "sm_tracks_runs: bad arguments");
  size_t at = 0;
  for (const Track& t : h->tracks) {
    std::memcpy(starts + at, t.starts.data(), t.starts.size() * sizeof(int64_t));
    std::memcpy(runs + at, t.runs.data(), t.runs.size() * sizeof(int64_t));
    at += t.starts.size();
  }
  return EMP_OK;
}

// ---- slice read-back (parity tests, save_panoptic) ----------------------------------------------------------------
int64_t emp_sm_slice_num_objects(const emp_stack_matcher* h, int64_t idx) {
  return (h && idx >= 0 && idx < (int64_t)h->stack.size()) ? (int64_t)h->stack[(size_t)idx].objs.size() : -1;
}

int emp_sm_slice_object_info(const emp_stack_matcher* h, int64_t idx, int64_t k, int64_t* label, int64_t* box4,
                             int64_t* n_runs) {
  EMP_REQUIRE(h && idx >= 0 && idx < (int64_t)h->stack.size() && k >= 0 && k < (int64_t)h->stack[(size_t)idx].objs.size(),
              "sm_slice_object_info: bad arguments");
  const FSlice& sl = h->stack[(size_t)idx];
  const FObj& o = sl.objs[(size_t)k];
  *label = o.label;
  std::memcpy(box4, o.box, sizeof(o.box));
  std::vector<int64_t> st, rn;
  object_runs(sl, o, st, rn);
  *n_runs = (int64_t)st.size();
  return EMP_OK;
}

int emp_sm_slice_object_runs(const emp_stack_matcher* h, int64_t idx, int64_t k, int64_t* starts, int64_t* runs) {
  EMP_REQUIRE(h && idx >= 0 && idx < (int64_t)h->stack.size() && k >= 0 && k < (int64_t)h->stack[(size_t)idx].objs.size(),
              "sm_slice_object_runs: bad arguments");
  const FSlice& sl = h->stack[(size_t)idx];
  std::vector<int64_t> st, rn;
  object_runs(sl, sl.objs[(size_t)k], st, rn);
  std::memcpy(starts, st.data(), st.size() * sizeof(int64_t));
  std::memcpy(runs, rn.data(), rn.size() * sizeof(int64_t));
  return EMP_OK;
}


// Segment gather for the rank-0 merge of per-slab partial trackers (multigpu.merge_partial_trackers): output segment j =
// n = cnt[j] runs copied from source array src_id[j] at offset src_off[j] to out + out_off[j], for the starts and the run
// lengths alike -- contiguous memcpys on worker threads instead of one numpy slice + concatenate per object and slab.
int emp_gather_segments_i64(const int64_t* const* h_src_a, const int64_t* const* h_src_b, const int32_t* h_src_id,
                            const int64_t* h_src_off, const int64_t* h_cnt, const int64_t* h_out_off, int64_t n_seg,
                            int64_t* h_out_a, int64_t* h_out_b) {
  EMP_REQUIRE(n_seg >= 0 && (n_seg == 0 || (h_src_a && h_src_b && h_src_id && h_src_off && h_cnt && h_out_off && h_out_a && h_out_b)),
              "gather_segments: null pointer");
  // memory-bound (and the destination is freshly allocated: first-touch page faults): more threads than the matcher's
  // default, contiguous ranges of segments per thread so that every thread writes its own stretch of the output
  const int hw = (int)std::thread::hardware_concurrency();
  const int T = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)std::max(sm_threads(), 8), (int64_t)(hw > 0 ? hw : 8), n_seg / 64 + 1}));
  auto work = [&](int t) {
    const int64_t j0 = n_seg * t / T, j1 = n_seg * (t + 1) / T;
    for (int64_t j = j0; j < j1; ++j) {
      const int64_t n = h_cnt[j];
      if (n <= 0) continue;
      std::memcpy(h_out_a + h_out_off[j], h_src_a[h_src_id[j]] + h_src_off[j], (size_t)n * sizeof(int64_t));
      std::memcpy(h_out_b + h_out_off[j], h_src_b[h_src_id[j]] + h_src_off[j], (size_t)n * sizeof(int64_t));
    }
  };
  if (T == 1) { work(0); return EMP_OK; }
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t) th.emplace_back(work, t);
  for (auto& x : th) x.join();
  return EMP_OK;
}

}  // extern "C"
