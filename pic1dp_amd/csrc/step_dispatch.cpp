// step_dispatch.cpp -- launch_step: the arguments of a whole-step launch in their device form, handed to the
// translation unit of the species' distribution (kernels_step.hip, one object per PIC1DP_STEP_DIST)
#include <algorithm>

#include "step_args.hpp"

namespace pic1dp {

hipError_t launch_step(const StepArgs &a, bool full, const LaunchCfg &lc, hipStream_t st) {
  StepArgsDev d{};
  d.x = a.x;
  d.v = a.v;
  d.w = a.w;
  d.p = a.p;
  d.E0 = a.E0;
  d.Eh = a.Eh;
  d.rho = a.rho;
  d.np = a.np;
  d.dt_half = a.dt_half;
  d.dt_full = a.dt_full;
  d.g = a.g;
  d.s = a.s;
  d.nt = a.stream_nt;
  d.t2 = a.t2;
  d.dg = a.dg;
  d.dist_out = a.dist_out;
  d.dist_partial = a.dist_partial;
  d.tabA = a.tabA;
  d.tabB = a.tabB;
  d.pred = a.pred;
  d.pred_nm = a.pred_kind == 2 ? (a.pred_private ? -2 : -1) : a.pred_nm;  // -1: k_step_sums, -2: k_step_one<PRIV>
  d.t2_mode = a.t2_mode;
  d.eh_re = a.eh_re;
  d.eh_im = a.eh_im;
  d.snx = a.g.dnx / a.g.lx;
  d.pred_k = a.dt_half * a.s.Z / a.s.m;
  d.fused = a.fused;
  d.tail = a.tail;
  d.dyn_tail = a.dyn_tail;
  d.fxb = a.fxb;
  {  // rows of pairs a workgroup takes (static grid stride; the drawn chunks are its own rows), two markers a pair
    const int64_t npair = a.np >> 1, stride = static_cast<int64_t>(lc.blocks) * lc.threads;
    // (at least 4096: a term times its scale then stays below 2^49, inside the 2^51 the conversion's magic number covers)
    d.fx_markers = std::max(4096.0, 2.0 * static_cast<double>((npair + stride - 1) / stride) * lc.threads + 2.0);
    d.fx_cap = 0x1p61 / d.fx_markers;
  }
  d.dscale = a.dscale;
  d.diag_fx = a.diag_fx;
  // full-f evaluates no f0 derivative: one instantiation (in the DIST 0 unit) serves every distribution
  if (!a.deltaf) return launch_step_dist<0>(d, a.deltaf, a.linear, full, lc, st);
  switch (a.iptcldist) {
    case 1: return launch_step_dist<1>(d, a.deltaf, a.linear, full, lc, st);
    case 2: return a.s.one_exp ? launch_step_dist<4>(d, a.deltaf, a.linear, full, lc, st)
                               : launch_step_dist<2>(d, a.deltaf, a.linear, full, lc, st);
    case 3: return a.s.one_exp ? launch_step_dist<5>(d, a.deltaf, a.linear, full, lc, st)
                               : launch_step_dist<3>(d, a.deltaf, a.linear, full, lc, st);
    default: return launch_step_dist<0>(d, a.deltaf, a.linear, full, lc, st);
  }
}


}  // namespace pic1dp
