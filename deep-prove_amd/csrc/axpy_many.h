// Host <-> device interface of Dev::axpy_many on the device (k_axpy_many / k_axpy_classes, kernels.inc): the descriptors the kernels read and the plan
// the host makes of a set of jobs. Kept apart from hip_dev.hip so that the kernel-emulation test (tests/support/kernel_emul) drives the kernel source
// with exactly the host code the product uses.
//
// acc[i] = init[i] (or 0) + sum_d x_d[i >> lg_rep_d] * coeff_d (pcs.h: the running oracle and the evaluation sum of a batch opening, basefold
// commit_phase.rs:187-359). The jobs SHORTER than the accumulator (33 of the 37 polynomials of a Dense-4M proof: each value repeated 2^lg_rep times)
// are first summed among their own length — class c = the jobs of one length n_c, out_c[i] = sum_d x_d[i] * coeff_d, one launch, blockIdx.y = class —
// and the pass over the accumulator then ADDS the class sums with their repetition (xext = 2: no multiplication). The multiplications run over
// sum_d |x_d| elements instead of nd * n_acc; field arithmetic is exact and every stored value canonical, so the accumulator is bit-identical either way.
#pragma once
#include "dev.h"
#include <map>
#include <vector>

namespace dp {

struct AxpyDesc { const void* x; int xext; unsigned lg_rep; size_t n_x; Ext coeff; };  // xext: 0 base field, 1 extension, 2 extension ADDED AS IT IS (coeff unused: a class sum)
struct AxpyClass { Ext* out; size_t n; int first, count; };                            // class sum `out` of length n over descriptors [first, first + count)
static_assert(sizeof(AxpyDesc) % 16 == 0 && sizeof(AxpyClass) % 8 == 0, "descriptor layout");

struct AxpyPlan {
  bool grouped = false;
  std::vector<AxpyDesc> members;      // grouped: the short jobs, class by class, each with lg_rep 0 (read by k_axpy_classes)
  std::vector<AxpyClass> classes;     // grouped: `out` is filled in by the caller (it owns the memory)
  std::vector<AxpyDesc> final_pass;   // descriptors of the pass over the accumulator; grouped: the full-length jobs, then one xext = 2 entry per class
  std::vector<size_t> class_final;    // grouped: index in final_pass of class c's entry (its `x` is the caller's to fill in)
  size_t max_class_n = 0;
  double member_bytes = 0, final_bytes = 0;  // algorithmic bytes of the two launches (without the accumulator itself)
};

// shapes are checked here: acc_n == x.n * rep, rep a power of two
inline AxpyPlan axpy_plan(const Dev::AxpyJob* jobs, size_t n, size_t acc_n, bool classes_enabled) {
  AxpyPlan p;
  auto fill = [](const Dev::AxpyJob& j, unsigned lg) { AxpyDesc d; d.x = j.x.p; d.xext = j.x.ext ? 1 : 0; d.lg_rep = lg; d.n_x = j.x.n; d.coeff = j.coeff; return d; };
  std::map<unsigned, std::vector<size_t>> by_rep;
  size_t nshort = 0;
  for (size_t i = 0; i < n; i++) {
    DP_REQUIRE(jobs[i].rep >= 1 && acc_n == jobs[i].x.n * jobs[i].rep && (jobs[i].rep & (jobs[i].rep - 1)) == 0, DP_ERR_SHAPE, "axpy_many: shapes");
    if (jobs[i].rep > 1) { by_rep[dp_ceil_log2(jobs[i].rep)].push_back(i); nshort++; }
  }
  p.grouped = classes_enabled && nshort >= 2;
  if (!p.grouped) {
    for (size_t i = 0; i < n; i++) { p.final_pass.push_back(fill(jobs[i], dp_ceil_log2(jobs[i].rep))); p.final_bytes += jobs[i].x.bytes(); }
    return p;
  }
  for (size_t i = 0; i < n; i++) if (jobs[i].rep == 1) { p.final_pass.push_back(fill(jobs[i], 0)); p.final_bytes += jobs[i].x.bytes(); }
  for (auto& kv : by_rep) {
    const size_t cn = acc_n >> kv.first;
    AxpyClass c; c.out = nullptr; c.n = cn; c.first = (int)p.members.size(); c.count = (int)kv.second.size();
    for (size_t i : kv.second) { p.members.push_back(fill(jobs[i], 0)); p.member_bytes += jobs[i].x.bytes(); }
    p.classes.push_back(c);
    AxpyDesc f; f.x = nullptr; f.xext = 2; f.lg_rep = kv.first; f.n_x = cn; f.coeff = ex_one();
    p.class_final.push_back(p.final_pass.size());
    p.final_pass.push_back(f);
    p.member_bytes += 16.0 * cn; p.final_bytes += 16.0 * cn;
    p.max_class_n = std::max(p.max_class_n, cn);
  }
  return p;
}

}  // namespace dp
