// quadtree.cpp -- host side of the extractor: ORBextractor::DistributeOctTree
// (reference src/vslam/src/ORBextractor.cc:587-811, DivideNode :529-585).
//
// The quadtree is inherently serial (each split decision depends on the running node count) and
// works on ~10^4 corners per level, so it stays on the host.  This is
// not a transliteration of the reference's std::list<ExtractorNode> of copied cv::KeyPoint
// vectors: nodes live in a pool, are linked through an intrusive doubly linked list that
// reproduces the reference's list order (children push_front'ed in n1..n4 order, parent erased),
// and each node owns a contiguous range of an index array that is stably partitioned in place,
// so no keypoint is ever copied.
//
// Tie-break: the reference sorts pair<int, ExtractorNode*> (ORBextractor.cc:732), i.e. equal
// sizes are ordered by POINTER VALUE, which is not deterministic.  Here (and in the oracle)
// equal sizes are ordered by node creation sequence, which is what a monotonically growing heap
// gives the reference in practice.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "quadtree.h"

namespace {
struct QNode {
  int ulx, uly, brx, bry;  // UL and BR corners (UR = (brx, uly), BL = (ulx, bry))
  int beg, end;            // range in idx[]
  int prev, next;          // intrusive list links (-1 = none)
  int seq;
  bool no_more;
};

struct QTree {
  std::vector<QNode> pool;
  std::vector<int> idx, tmp;
  int head = -1, tail = -1, count = 0, seq = 0;
  const float *x, *y;

  int alloc(const QNode& n) {
    pool.push_back(n);
    return (int)pool.size() - 1;
  }
  void push_back(int id) {
    QNode& n = pool[id];
    n.prev = tail; n.next = -1;
    if (tail >= 0) pool[tail].next = id; else head = id;
    tail = id;
    ++count;
  }
  void push_front(int id) {
    QNode& n = pool[id];
    n.next = head; n.prev = -1;
    if (head >= 0) pool[head].prev = id; else tail = id;
    head = id;
    ++count;
  }
  int erase(int id) {  // returns next
    QNode& n = pool[id];
    const int nx = n.next;
    if (n.prev >= 0) pool[n.prev].next = n.next; else head = n.next;
    if (n.next >= 0) pool[n.next].prev = n.prev; else tail = n.prev;
    --count;
    return nx;
  }

  // DivideNode (:529-585): children geometry + stable 4-way partition of the parent's key range.
  // child order n1 (UL), n2 (UR), n3 (BL), n4 (BR); returns their [beg,end) and boxes.
  void divide(int id, QNode out[4]) {
    const QNode p = pool[id];
    const int halfX = (int)std::ceil(static_cast<float>(p.brx - p.ulx) / 2);
    const int halfY = (int)std::ceil(static_cast<float>(p.bry - p.uly) / 2);
    const int mx = p.ulx + halfX, my = p.uly + halfY;
    int cnt[4] = {0, 0, 0, 0};
    for (int i = p.beg; i < p.end; ++i) {
      const int k = idx[i];
      const int q = (x[k] < mx) ? ((y[k] < my) ? 0 : 2) : ((y[k] < my) ? 1 : 3);
      ++cnt[q];
    }
    int off[4];
    off[0] = p.beg;
    for (int q = 1; q < 4; ++q) off[q] = off[q - 1] + cnt[q - 1];
    int cur[4] = {off[0], off[1], off[2], off[3]};
    for (int i = p.beg; i < p.end; ++i) {
      const int k = idx[i];
      const int q = (x[k] < mx) ? ((y[k] < my) ? 0 : 2) : ((y[k] < my) ? 1 : 3);
      tmp[cur[q]++] = k;
    }
    std::copy(tmp.begin() + p.beg, tmp.begin() + p.end, idx.begin() + p.beg);
    const int bx[4][4] = {{p.ulx, p.uly, mx, my}, {mx, p.uly, p.brx, my}, {p.ulx, my, mx, p.bry}, {mx, my, p.brx, p.bry}};
    for (int q = 0; q < 4; ++q) {
      out[q].ulx = bx[q][0]; out[q].uly = bx[q][1]; out[q].brx = bx[q][2]; out[q].bry = bx[q][3];
      out[q].beg = off[q]; out[q].end = off[q] + cnt[q];
      out[q].no_more = (cnt[q] == 1);
      out[q].prev = out[q].next = -1;
    }
  }
};
}  // namespace

void asd_distribute_octtree(const float* x, const float* y, const float* response, int n, int minX, int maxX,
                            int minY, int maxY, int N, std::vector<int>& selected) {
  selected.clear();
  if (n <= 0) return;
  const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
  if (nIni < 1) return;
  const float hX = static_cast<float>(maxX - minX) / nIni;
  QTree t;
  t.x = x; t.y = y;
  t.pool.reserve(4 * (size_t)std::max(N, 64) + 64);
  t.idx.resize(n);
  t.tmp.resize(n);
  // root assignment vpIniNodes[kp.pt.x/hX] (:617): stable counting sort by root index
  std::vector<int> root_of(n), rcount(nIni + 1, 0);
  for (int i = 0; i < n; ++i) {
    int r = (int)(x[i] / hX);
    if (r >= nIni) r = nIni - 1;
    root_of[i] = r;
    ++rcount[r + 1];
  }
  for (int r = 0; r < nIni; ++r) rcount[r + 1] += rcount[r];
  {
    std::vector<int> cur(rcount.begin(), rcount.end() - 1);
    for (int i = 0; i < n; ++i) t.idx[cur[root_of[i]]++] = i;
  }
  for (int r = 0; r < nIni; ++r) {
    QNode nd{};
    nd.ulx = (int)(hX * static_cast<float>(r)); nd.uly = 0;
    nd.brx = (int)(hX * static_cast<float>(r + 1)); nd.bry = maxY - minY;
    nd.beg = rcount[r]; nd.end = rcount[r + 1];
    nd.seq = t.seq++;
    nd.no_more = false;
    t.push_back(t.alloc(nd));
  }
  for (int id = t.head; id >= 0;) {
    QNode& nd = t.pool[id];
    const int sz = nd.end - nd.beg;
    if (sz == 1) { nd.no_more = true; id = nd.next; }
    else if (sz == 0) id = t.erase(id);
    else id = nd.next;
  }
  struct SP { int size, seq, id; };
  std::vector<SP> expand, prev_expand;
  auto add_children = [&](QNode ch[4], int* n_to_expand) {
    for (int q = 0; q < 4; ++q) {
      const int sz = ch[q].end - ch[q].beg;
      if (sz > 0) {
        ch[q].seq = t.seq++;
        const int cid = t.alloc(ch[q]);
        t.push_front(cid);
        if (sz > 1) {
          if (n_to_expand) ++*n_to_expand;
          expand.push_back(SP{sz, ch[q].seq, cid});
        }
      }
    }
  };
  bool finish = false;
  QNode ch[4];
  while (!finish) {
    int prev_size = t.count;
    int n_to_expand = 0;
    expand.clear();
    for (int id = t.head; id >= 0;) {
      if (t.pool[id].no_more) { id = t.pool[id].next; continue; }
      t.divide(id, ch);
      add_children(ch, &n_to_expand);
      id = t.erase(id);
    }
    if (t.count >= N || t.count == prev_size) {
      finish = true;
    } else if (t.count + n_to_expand * 3 > N) {
      while (!finish) {
        prev_size = t.count;
        prev_expand.swap(expand);
        expand.clear();
        std::sort(prev_expand.begin(), prev_expand.end(),
                  [](const SP& a, const SP& b) { return a.size != b.size ? a.size < b.size : a.seq < b.seq; });
        for (int j = (int)prev_expand.size() - 1; j >= 0; --j) {
          t.divide(prev_expand[j].id, ch);
          add_children(ch, nullptr);
          t.erase(prev_expand[j].id);
          if (t.count >= N) break;
        }
        if (t.count >= N || t.count == prev_size) finish = true;
      }
    }
  }
  selected.reserve(t.count);
  for (int id = t.head; id >= 0; id = t.pool[id].next) {
    const QNode& nd = t.pool[id];
    int best = t.idx[nd.beg];
    float max_resp = response[best];
    for (int i = nd.beg + 1; i < nd.end; ++i) {
      const int k = t.idx[i];
      if (response[k] > max_resp) { best = k; max_resp = response[k]; }
    }
    selected.push_back(best);
  }
}
