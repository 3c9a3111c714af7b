// Solver kernel choice (host code): ONE function decides which patch-solve kernel a plan uses,
// with which LDS size and which stages fused in.  slod_plan_create calls it (and rejects the plan
// when nothing fits), slod_plan_execute launches exactly that choice.
#include "slod_device.h"

#include <cstdlib>
#include <cstring>

// Tuning knobs of the experiments in tools/ and of the kernel-variant parity tests; read once
// per plan (slod_plan_create), never in the launch path.
//   SLOD_SOLVE=mf|tw|ws|coop  force a kernel family       SLOD_FUSE_SELECT=0  selection as its own launch
//   SLOD_FUSE_ASSEMBLE=0      stencil assembly as its own launch
//   SLOD_FUSE_M=1 (ws only)   SLOD_TWISTED=0|1 (coop only) SLOD_DEBUG=1 print the choice
//   SLOD_BALANCE=0            launch the patches in the caller's order (default: balanced over the CUs)
SlodTuning slod_read_tuning()
{
  SlodTuning t;
  if (const char *sel = getenv("SLOD_SOLVE"))
    t.solver = !strcmp(sel, "mf") ? SLOD_K_MF : !strcmp(sel, "tw") ? SLOD_K_TW : !strcmp(sel, "ws") ? SLOD_K_WS :
               !strcmp(sel, "coop") ? SLOD_K_COOP : !strcmp(sel, "nd") ? SLOD_K_ND : 0;
  if (const char *e = getenv("SLOD_FUSE_SELECT"))
    t.fuse_select = atoi(e) ? 1 : 0;
  if (const char *e = getenv("SLOD_FUSE_ASSEMBLE"))
    t.fuse_assemble = atoi(e) ? 1 : 0;
  if (const char *e = getenv("SLOD_FUSE_M"))
    t.fuse_m = atoi(e) ? 1 : 0;
  if (const char *e = getenv("SLOD_TWISTED"))
    t.twisted = atoi(e) ? 1 : 0;
  if (const char *e = getenv("SLOD_BALANCE"))
    t.balance = atoi(e) ? 1 : 0;
  if (const char *e = getenv("SLOD_DEBUG"))
    t.debug = atoi(e) ? 1 : 0;
  return t;
}

bool slod_choose_solver(int S, int n_sub, int m_max, int L_max, int nc_max, int nb_buf, int nf_max, size_t n_patches,
                        const SlodTuning &t_in, SlodSolveChoice *out)
{
  const size_t    lds_max = 160 * 1024;
  SlodSolveChoice c;
  const int       nd_nv = slod_solve_nd_cell(S, n_sub, m_max, L_max);
  SlodTuning      t     = t_in;
  if (t.solver == SLOD_K_ND && nd_nv == 0)
    t.solver = 0; // nested dissection does not take this plan (vector problem, cell size): automatic choice
  c.debug = t.debug;
  // Kernel families:
  //   tw   twisted + wave-specialised VALU Gauss-Jordan (default: fastest on every BASELINE size,
  //        tools/solver_compare.py)
  //   mf   MFMA-factorised: blocked Gauss-Jordan on the fp64 matrix pipe, twisted, column-tile
  //        private right-hand-side streams (SLOD_SOLVE=mf; automatic only where tw does not fit)
  //   ws   wave-specialised, one chain
  //   coop all threads cooperate on every pivot (also for tiles narrower than the band)
  const int  wt = slod_solve_ws_tile(m_max);
  const bool ws_fits = wt > 0 && wt >= 2 * S - 1;
  const auto want = [&](int k) { return t.solver == 0 || t.solver == k; };
  const size_t lds_sel = slod_select_lds_bytes(S, nb_buf, nc_max, nf_max);
  const bool   mf_fits = slod_solve_mf_tiles(S, m_max) > 0 && slod_solve_mf_lds_bytes(S, m_max, nc_max) <= lds_max;
  const bool   tw_fits = ws_fits && slod_solve_tw_lds_bytes(S, m_max, nc_max) <= lds_max;
  if (nd_nv > 0 && t.solver == SLOD_K_ND)
    {
      //   nd   nested dissection: static condensation per cell, edge sets, skeleton lines
      c.kind          = SLOD_K_ND;
      c.nv            = nd_nv;
      c.lds           = slod_solve_nd_lds_bytes(nd_nv, m_max, nc_max);
      c.v_patch_elems = slod_solve_nd_scratch(nd_nv, m_max, L_max, nc_max);
      c.fuse_assemble = t.fuse_assemble ? 1 : 0;
      c.fuse_select   = 0; // 512-thread workgroups: the selection stage is its own launch
    }
  else if (mf_fits && (t.solver == SLOD_K_MF || (t.solver == 0 && !tw_fits && !ws_fits)))
    {
      c.kind          = SLOD_K_MF;
      c.lds           = slod_solve_mf_lds_bytes(S, m_max, nc_max);
      c.v_line_pad    = 16 * slod_solve_mf_tiles(S, m_max);
      c.v_line_elems  = (size_t)c.v_line_pad * c.v_line_pad;
      c.fuse_assemble = (S == 1 && t.fuse_assemble) ? 1 : 0;
      c.fuse_select   = (S == 1 && t.fuse_select && lds_sel <= lds_max) ? 1 : 0;
      if (c.fuse_select && lds_sel > c.lds)
        c.lds = lds_sel;
    }
  else if (want(SLOD_K_TW) && tw_fits)
    {
      c.kind          = SLOD_K_TW;
      c.lds           = slod_solve_tw_lds_bytes(S, m_max, nc_max);
      c.v_line_pad    = 8 * wt;
      c.v_line_elems  = (size_t)64 * wt * wt; // the 8 x 8 lane tiles, both triangles
      c.fuse_assemble = (S == 1 && t.fuse_assemble) ? 1 : 0;
      // the selection stage runs in the same launch (scalar problems) while four workgroups
      // still fit a CU
      c.fuse_select = (S == 1 && t.fuse_select && lds_sel <= 64 * 1024) ? 1 : 0;
      if (c.fuse_select && lds_sel > c.lds)
        c.lds = lds_sel;
    }
  else if ((want(SLOD_K_WS) || t.solver == SLOD_K_TW) && ws_fits &&
           slod_solve_ws_lds_bytes(S, m_max, nc_max) <= lds_max)
    {
      c.kind       = SLOD_K_WS;
      c.lds        = slod_solve_ws_lds_bytes(S, m_max, nc_max);
      c.v_line_pad = 8 * wt;
      c.v_line_elems = (size_t)c.v_line_pad * c.v_line_pad;
      // fusing M = sum_l R_l^T Z_l into the helper waves saves the selection stage's re-read of X
      // but costs a fourth barrier per line; measured neutral on C2, so opt-in (SLOD_FUSE_M=1)
      c.m_fused = (t.fuse_m && nc_max * nc_max <= 192 * 4) ? 1 : 0;
    }
  else if (t.solver == 0 || t.solver == SLOD_K_COOP || !ws_fits)
    {
      // twisted (two chains, 512 threads) when the GPU is not full anyway: it halves the
      // dependent chain per patch; one chain per patch otherwise
      int tw = t.twisted >= 0 ? t.twisted : (n_patches < 3 * 256 ? 1 : 0);
      if (slod_solve_lds_bytes(S, m_max, nc_max, tw) > lds_max)
        tw = 0;
      if (slod_solve_lds_bytes(S, m_max, nc_max, tw) > lds_max || (m_max + 15) / 16 > 7)
        return false;
      c.kind       = SLOD_K_COOP;
      c.twisted    = tw;
      c.lds        = slod_solve_lds_bytes(S, m_max, nc_max, tw);
      c.v_line_pad = m_max;
      c.v_line_elems = (size_t)m_max * m_max;
    }
  else
    return false;
  if (!c.fuse_select && lds_sel > lds_max)
    return false; // the stand-alone selection launch does not fit either
  *out = c;
  return true;
}

hipError_t slod_launch_solve(int S, const SlodSolveChoice &c, SlodKernelArgs &a, int n_patches, hipStream_t st)
{
  a.m_fused       = c.m_fused;
  a.fuse_select   = c.fuse_select;
  a.fuse_assemble = c.fuse_assemble;
  a.debug         = c.debug;
  switch (c.kind)
    {
      case SLOD_K_MF:
        return slod_launch_solve_mf(S, a, n_patches, c.lds, st);
      case SLOD_K_TW:
        return slod_launch_solve_tw(S, a, n_patches, c.lds, st);
      case SLOD_K_WS:
        return slod_launch_solve_ws(S, a, n_patches, c.lds, st);
      case SLOD_K_COOP:
        return slod_launch_solve_coop(S, c.twisted, a, n_patches, st);
      case SLOD_K_ND:
        a.nv = c.nv;
        return slod_launch_solve_nd(a, n_patches, c.lds, st);
      default:
        return hipErrorInvalidValue;
    }
}
