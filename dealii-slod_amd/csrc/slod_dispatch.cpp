// Solver kernel choice (host code).
#include "slod_device.h"

#include <cstdlib>
#include <cstring>

// true when slod_launch_solve will pick k_solve_tw for a scalar problem: that kernel then assembles
// the stencil itself (SLOD_FUSE_ASSEMBLE=0 keeps the separate k_assemble launch)
bool slod_solve_fuses_assemble(int S, const SlodKernelArgs &a)
{
  const char *sel = getenv("SLOD_SOLVE");
  const char *fa  = getenv("SLOD_FUSE_ASSEMBLE");
  const bool  fits = slod_solve_ws_tile(a.m_max) >= 2 * S - 1 && slod_solve_ws_tile(a.m_max) > 0;
  return S == 1 && !(fa && !atoi(fa)) && fits && (!sel || !strcmp(sel, "tw")) &&
         slod_solve_tw_lds_bytes(S, a.m_max, a.nc_max) <= 160 * 1024;
}

hipError_t slod_launch_solve(int S, SlodKernelArgs &a, int n_patches, hipStream_t st)
{
  a.m_fused = 0;
  a.fuse_select = 0;
  // Kernel choice (SLOD_SOLVE=tw|ws|coop forces one):
  //   tw   twisted + wave-specialised (default): two GJ waves + two helper waves per patch
  //   ws   wave-specialised, one chain: one GJ wave + three helpers (keeps V, Z in LDS)
  //   coop all threads cooperate on every pivot (k_solve, also for tiles narrower than the band)
  {
    const char *sel  = getenv("SLOD_SOLVE");
    const bool  fits = slod_solve_ws_tile(a.m_max) >= 2 * S - 1 && slod_solve_ws_tile(a.m_max) > 0;
    const bool  want_tw = !sel || !strcmp(sel, "tw"), want_ws = !sel || !strcmp(sel, "ws") || !strcmp(sel, "tw");
    if (fits && want_tw && slod_solve_tw_lds_bytes(S, a.m_max, a.nc_max) <= 160 * 1024)
      {
        size_t lds = slod_solve_tw_lds_bytes(S, a.m_max, a.nc_max);
        // the selection stage runs in the same launch (scalar problems; SLOD_FUSE_SELECT=0 splits it off)
        const char  *fs   = getenv("SLOD_FUSE_SELECT");
        const size_t lds2 = slod_select_lds_bytes(S, a.nb_buf, a.nc_max, a.nf_max);
        a.fuse_select     = (S == 1 && !(fs && !atoi(fs)) && lds2 <= 64 * 1024) ? 1 : 0;
        if (a.fuse_select && lds2 > lds)
          lds = lds2;
        return slod_launch_solve_tw(S, a, n_patches, lds, st);
      }
    if (fits && want_ws && slod_solve_ws_lds_bytes(S, a.m_max, a.nc_max) <= 160 * 1024)
      {
        const size_t lds = slod_solve_ws_lds_bytes(S, a.m_max, a.nc_max);
        // fusing M = sum_l R_l^T Z_l into the helper waves saves k_select's re-read of X but
        // costs a fourth barrier per line; measured neutral on C2, so opt-in (SLOD_FUSE_M=1)
        const char *fm = getenv("SLOD_FUSE_M");
        a.m_fused      = (fm && atoi(fm) && a.nc_max * a.nc_max <= 192 * 4) ? 1 : 0;
        return slod_launch_solve_ws(S, a, n_patches, lds, st);
      }
  }
  // twisted (two chains, 512 threads) when the GPU is not full anyway: it halves the
  // dependent chain per patch; one chain per patch otherwise (same work, more patches
  // resident).  SLOD_TWISTED=0/1 overrides.
  int tw = n_patches < 3 * 256 ? 1 : 0;
  if (const char *env = getenv("SLOD_TWISTED"))
    tw = atoi(env) ? 1 : 0;
  if (slod_solve_lds_bytes(S, a.m_max, a.nc_max, tw) > 160 * 1024)
    tw = 0;
  return slod_launch_solve_coop(S, tw, a, n_patches, st);
}
