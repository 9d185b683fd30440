// ORACLE (test infrastructure only).
// Restates multilinear_extensions/src/mle.rs (DenseMultilinearExtension, fix_variables*, fix_high_variables*,
// evaluate) and virtual_poly.rs (VirtualPolynomial, build_eq_x_r_vec, eq_eval), plus zkml/src/commit/mod.rs
// (compute_betas_eval, identity_eval). Index convention: little endian, 0b1011 -> P(1,1,0,1) (mle.rs:226-228).
#pragma once
#include "field.hpp"
#include "par.hpp"
#include <memory>
#include <utility>

namespace orc {

// FieldType::{Base,Ext} + num_vars  (mle.rs:137-181)
struct Mle {
  bool is_ext = false;
  std::vector<u64> b;
  std::vector<E> e;
  unsigned nv = 0;
  size_t len() const { return is_ext ? e.size() : b.size(); }
  E at(size_t i) const { return is_ext ? e[i] : e_from(b[i]); }
  static Mle from_base(std::vector<u64> v) {
    Mle m; m.is_ext = false; m.nv = log2_strict(v.size()); m.b = std::move(v); return m;
  }
  static Mle from_ext(std::vector<E> v) {
    Mle m; m.is_ext = true; m.nv = log2_strict(v.size()); m.e = std::move(v); return m;
  }
  static Mle from_i64(const std::vector<int64_t>& v) {
    std::vector<u64> b(v.size());
    for (size_t i = 0; i < v.size(); i++) b[i] = orc::from_i64(v[i]);
    return from_base(std::move(b));
  }
  // fix_variables / fix_variables_in_place (mle.rs:454-525): out[i] = a[2i] + r*(a[2i+1]-a[2i])
  void fix_low_in_place(E r) {
    assert(nv > 0);
    size_t n = len();
    std::vector<E> out(n / 2);
    par_for(n / 2, [&](size_t lo, size_t hi) {
      if (is_ext) {
        for (size_t i = lo; i < hi; i++) out[i] = eadd(e[2 * i], emul(esub(e[2 * i + 1], e[2 * i]), r));
      } else {
        for (size_t i = lo; i < hi; i++) out[i] = eadd(emul_base(r, fsub(b[2 * i + 1], b[2 * i])), e_from(b[2 * i]));
      }
    });
    e = std::move(out);
    b.clear();
    is_ext = true;
    nv -= 1;
  }
  void fix_low_in_place(const std::vector<E>& pt) { for (E r : pt) fix_low_in_place(r); }
  // fix_high_variables_in_place (mle.rs:562-603): for t = k-1..0: lo[i] += q[t]*(hi[i]-lo[i])
  void fix_high_in_place(const std::vector<E>& pt) {
    assert(pt.size() <= nv);
    for (size_t t = pt.size(); t-- > 0;) {
      E r = pt[t];
      size_t half = len() / 2;
      std::vector<E> out(half);
      par_for(half, [&](size_t lo, size_t hi) {
        if (is_ext) {
          for (size_t i = lo; i < hi; i++) out[i] = eadd(e[i], emul(esub(e[i + half], e[i]), r));
        } else {
          for (size_t i = lo; i < hi; i++) out[i] = eadd(emul_base(r, fsub(b[i + half], b[i])), e_from(b[i]));
        }
      });
      e = std::move(out);
      b.clear();
      is_ext = true;
      nv -= 1;
    }
  }
  // evaluate (mle.rs:607-623)
  E evaluate(const std::vector<E>& pt) const {
    if (pt.size() != nv) throw std::runtime_error("MLE size does not match the point");
    Mle m = *this;
    m.fix_low_in_place(pt);
    return m.at(0);
  }
};
using MleP = std::shared_ptr<Mle>;
static inline MleP mk(Mle m) { return std::make_shared<Mle>(std::move(m)); }

// build_eq_x_r_vec (virtual_poly.rs:370-453): index bit t <-> r[t]
static inline std::vector<E> build_eq_x_r_vec(const std::vector<E>& r) {
  std::vector<E> buf(size_t(1) << r.size());
  buf[0] = e_one();
  size_t cur = 1;
  for (size_t t = r.size(); t-- > 0;) {
    for (size_t j = cur; j-- > 0;) {
      E prod = emul(r[t], buf[j]);
      buf[2 * j + 1] = prod;
      buf[2 * j] = esub(buf[j], prod);
    }
    cur *= 2;
  }
  return buf;
}
// compute_betas_eval (zkml/src/commit/mod.rs:10-28): same table, serial DP
static inline std::vector<E> compute_betas_eval(const std::vector<E>& r) {
  size_t n = r.size();
  std::vector<E> betas(size_t(1) << n, e_zero());
  betas[0] = e_one();
  for (size_t i = 0; i < n; i++) {
    size_t cs = size_t(1) << i;
    std::vector<E> tmp(betas.begin(), betas.begin() + cs);
    E re = r[n - 1 - i];
    for (size_t j = 0; j < cs; j++) {
      E t = emul(re, tmp[j]);
      betas[2 * j] = esub(tmp[j], t);
      betas[2 * j + 1] = t;
    }
  }
  return betas;
}
// eq_eval (virtual_poly.rs:308-322)
static inline E eq_eval(const std::vector<E>& x, const std::vector<E>& y) {
  if (x.size() != y.size()) throw std::runtime_error("eq_eval: length mismatch");
  E res = e_one();
  for (size_t i = 0; i < x.size(); i++) {
    E xy = emul(x[i], y[i]);
    res = emul(res, eadd(esub(esub(eadd(xy, xy), x[i]), y[i]), e_one()));
  }
  return res;
}
// identity_eval (zkml/src/commit/mod.rs:41-53): on the min length
static inline E identity_eval(const std::vector<E>& r1, const std::vector<E>& r2) {
  size_t n = std::min(r1.size(), r2.size());
  E ev = e_one();
  for (size_t i = 0; i < n; i++)
    ev = emul(ev, eadd(emul(r1[i], r2[i]), emul(esub(e_one(), r1[i]), esub(e_one(), r2[i]))));
  return ev;
}

// VirtualPolynomial (virtual_poly.rs:50-60,147-180): sum_i c_i * prod_j MLE; MLEs de-duplicated by pointer.
struct VirtualPolynomial {
  unsigned max_degree = 0, max_num_variables = 0;
  std::vector<std::pair<E, std::vector<size_t>>> products;
  std::vector<MleP> flattened;
  explicit VirtualPolynomial(unsigned nv) : max_num_variables(nv) {}
  void add_mle_list(const std::vector<MleP>& list, E coeff) {
    assert(!list.empty());
    for (auto& m : list) { assert(m->nv <= max_num_variables); assert(m->nv == list[0]->nv); }
    max_degree = std::max<unsigned>(max_degree, list.size());
    std::vector<size_t> idx;
    for (auto& m : list) {
      size_t k = 0;
      for (; k < flattened.size(); k++) if (flattened[k].get() == m.get()) break;
      if (k == flattened.size()) flattened.push_back(m);
      idx.push_back(k);
    }
    products.push_back({coeff, idx});
  }
};

}  // namespace orc
