// tools/scan_bench.hip -- stand-alone timing + bit-exactness harness for the approximate-search scan (K5w).
// Random band-major dB planes and a random sync table of the real shape; every variant must reproduce the generic
// kernel (K5, one 4-byte gather per term) bit for bit.   build: make -C audiowmark_amd/csrc scan_bench
#include "../audiowmark_amd/csrc/hip/kernels.hh"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf (stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString (e_)); return 1; } } while (0)

int
main (int argc, char **argv)
{
  setvbuf (stdout, nullptr, _IONBF, 0);
  const long long n_db = argc > 1 && atoll (argv[1]) > 0 ? atoll (argv[1]) : 55500;      // ~21.5 min of frames
  const int reps = argc > 2 ? atoi (argv[2]) : 10;
  // argv[3] = "clip": the CLIP search of padded 30 s clips -- 6692 frames per plane, two blocks (R = 170 rows per bit over 4452
  // frames), only the frames [3170, 4462) transformed (have = 1), zeros elsewhere; 256 planes = 64 clips x 4 shifts
  const bool clip = argc > 3 && !strcmp (argv[3], "clip");
  const int R = clip ? 170 : 85, total = clip ? 4452 : 2226, n_planes = clip ? 256 : 4;
  std::mt19937_64 rng (42);
  std::vector<int> packed (6 * R * 64, 0);
  for (int bit = 0; bit < 6; bit++)
    {
      std::vector<int> frames (total);
      for (int i = 0; i < total; i++) frames[i] = i;
      std::shuffle (frames.begin(), frames.end(), rng);
      frames.resize (R);
      std::sort (frames.begin(), frames.end());
      for (int r = 0; r < R; r++)
        {
          int *row = &packed[(bit * R + r) * 64];
          std::vector<int> bands (81);
          for (int i = 0; i < 81; i++) bands[i] = i;
          std::shuffle (bands.begin(), bands.end(), rng);
          std::sort (bands.begin(), bands.begin() + 30);
          std::sort (bands.begin() + 30, bands.begin() + 60);
          for (int i = 0; i < 60; i++) row[i] = bands[i];
          row[60] = frames[r];
          row[61] = r + 1 < R ? frames[r + 1] : 0x7fffffff;
        }
    }
  const long long S = n_db - total, ld = (n_db + 63) & ~63LL, plane = ld * 81, q_stride = (S + 63) & ~63LL;
  std::vector<float> db (n_planes * plane);
  std::uniform_real_distribution<float> dist (-70.f, -5.f);
  for (auto& v : db) v = dist (rng);
  std::vector<char> have (size_t (n_planes) * ld, 0);
  const long long run0 = argc > 4 ? atoll (argv[4]) : 3170, run1 = argc > 5 ? atoll (argv[5]) : 4462;
  if (clip)
    for (int p = 0; p < n_planes; p++)
      for (long long f = 0; f < ld; f++)
        {
          const bool in_run = f >= run0 + (p & 3) && f < run1 + (p % 5);       // (not the same run in every plane)
          have[p * ld + f] = in_run && f < n_db;
          if (!in_run)
            for (int b = 0; b < 81; b++)
              db[p * plane + b * ld + f] = 0.f;
        }
  char *d_have = nullptr;
  if (clip)
    {
      CK (hipMalloc (&d_have, have.size()));
      CK (hipMemcpy (d_have, have.data(), have.size(), hipMemcpyHostToDevice));
    }
  std::vector<unsigned> chains (12 * R * 8);
  awmk::pack_scan_chains (packed.data(), R, chains.data());
  unsigned *d_chains;
  CK (hipMalloc (&d_chains, chains.size() * 4));
  CK (hipMemcpy (d_chains, chains.data(), chains.size() * 4, hipMemcpyHostToDevice));
  float *d_db; int *d_tab; double *d_q;
  CK (hipMalloc (&d_db, db.size() * 4));
  CK (hipMalloc (&d_tab, packed.size() * 4));
  CK (hipMalloc (&d_q, n_planes * q_stride * 8));
  CK (hipMemcpy (d_db, db.data(), db.size() * 4, hipMemcpyHostToDevice));
  CK (hipMemcpy (d_tab, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
  awmk::SyncScanArgs sa {};
  sa.db = d_db; sa.plane_stride = plane; sa.have_plane_stride = ld; sa.row_stride = 1; sa.band_stride = ld; sa.have_row_stride = 1;
  sa.n_lanes = S; sa.n_planes = n_planes; sa.min_delta = 0.01; sa.quality = d_q; sa.q_stride = q_stride;
  sa.table.packed = d_tab; sa.table.rows_per_bit = R; sa.table.chains = d_chains;
  sa.have = d_have;
  hipStream_t st; CK (hipStreamCreate (&st));
  hipEvent_t e0, e1; CK (hipEventCreate (&e0)); CK (hipEventCreate (&e1));
  std::vector<double> ref (n_planes * q_stride), got (n_planes * q_stride);
  const double cand = double (n_planes) * S;
  auto run = [&] (const char *name, int variant, std::vector<double>& out) -> int {
    CK (hipMemsetAsync (d_q, 0xff, n_planes * q_stride * 8, st));
    auto launch = [&] () -> hipError_t {
      switch (variant)
        {
        case 0:  return awmk::launch_sync_scan (st, sa);
        default:
          {
            awmk::SyncScanArgs w = sa;
            w.have_is_run = 1;
            return awmk::launch_sync_scan_window (st, w, total);
          }
        }
    };
    CK (launch());
    CK (hipStreamSynchronize (st));
    CK (hipMemcpy (out.data(), d_q, out.size() * 8, hipMemcpyDeviceToHost));
    CK (hipEventRecord (e0, st));
    for (int i = 0; i < reps; i++) CK (launch());
    CK (hipEventRecord (e1, st));
    CK (hipEventSynchronize (e1));
    float ms; CK (hipEventElapsedTime (&ms, e0, e1));
    ms /= reps;
    long long bad = 0;
    if (&out != &ref)
      for (int p = 0; p < n_planes; p++)
        for (long long c = 0; c < S; c++)
          bad += memcmp (&out[p * q_stride + c], &ref[p * q_stride + c], 8) != 0;
    printf ("%-28s %8.4f ms  %7.2f TB/s of gathered terms  mismatches %lld / %.0f\n", name, ms, cand * 30600 * 4 / (ms * 1e-3) / 1e12, bad, cand);
    return 0;
  };
  if (run ("generic K5 (4-byte gathers, L2)", 0, ref)) return 1;
  if (run ("K5w (LDS ring, quads, loader wave)", 1, got)) return 1;
  return 0;
}
