// scan.hip -- K5w: SyncFinder::sync_decode for every start frame of the approximate search
// (reference syncfinder.cc:116-153 evaluated by search_approx, syncfinder.cc:171-256).
//
// Work: a candidate start frame c sums, for each of the 6 sync bits, the dB values of 30 "up" and 30 "down" bands of the
// 85 (BLOCK) sync frames of that bit: umag += db[c + frame][up[i]], dmag += db[c + frame][down[i]] -- 12 chains of 2550
// float additions per candidate whose ORDER is part of the result (sync qualities decide which positions are refined).
// The only freedom is across candidates and across chains.
//
// Gathers.  The planes are band-major (db[band][frame], one plane per 256-sample shift), so the candidates c, c+1, ... of
// one (sync frame, band) term are adjacent floats.  A lane owns FOUR adjacent candidates and fetches a term for all of them
// with ONE ds_read_b128 from a ring of the matrix in LDS (256 B/clk/CU instead of the 128 B/clk of 4-byte gathers,
// MI355X_MICROARCH.md, LDS table).  A 16-byte read must be 16-byte aligned, but the quad a lane needs starts at
// frame + 4 lane, i.e. at any alignment j = frame mod 4: the lane reads the aligned quad that starts j floats earlier
// (candidates 4 lane - j .. 4 lane - j + 3) and takes the elements that belong to its upper candidates from the lane
// above through the DPP operand of the addition itself (v_add_f32_dpp wave_shl:1, no extra instruction).  Lane 63 has no
// lane above: a wave delivers 252 of its 256 candidates, tiles advance by 252.
//
// Workgroup = one tile of 252 candidates: 12 CHAIN waves (sync bit x up / down; 4 accumulators per lane, 30 gathers + 120
// additions per sync frame) and one LOADER wave.  There is no workgroup barrier in the steady state:
//   * Ring: 7 chunks of 64 frames, chunk-major: s_win[slot][band (84, 81 used)][64 frames] = 150.5 KB.  One LDS-DMA
//     instruction (global_load_lds_dwordx4: every lane names its own 16 source bytes, the destination is wave base +
//     16 lane) moves 4 bands x 64 frames = 1 KiB of a chunk, 21 of them a whole chunk, without passing through registers.
//   * The loader keeps two chunks in flight, publishes `loaded` when a chunk has landed, and issues the next one as soon as
//     every chain has moved past the frames it replaces.  A chain checks before a sync frame that its 256 + 3 frames are
//     resident and publishes the frame of its next row afterwards.  The six bits' sync frames are unevenly spread over the
//     block; with a barrier per ring step the workgroup waited for the slowest bit a third of its time (round 1 kernel).
//   * The row descriptor of the NEXT sync frame (30 band bytes + next frame: one 32-byte scalar load) is requested before
//     the gathers of the current one.
//
// Measured on MI355X (tools/scan_bench, 55 500 frames x 4 shifts = 213 k candidates, bit-identical qualities):
//   generic K5 (4-byte gathers from L2)               1.63 ms
//   round 1 K5w (ring + barriers, ds_read_b32)        0.62 ms
//   quads + DPP, still one barrier pair per ring step 0.61 ms   (the gathers were never the limit)
//   this kernel                                       0.37 ms   = 70 TB/s of gathered terms of the ~150 TB/s the LDS delivers at 256 B/clk/CU
//                                                                 (effective clock under this kernel 2.3 GHz, GRBM_GUI_ACTIVE; an earlier
//                                                                 "1.6 GHz" came from s_memtime, which counts a constant reference clock)
#include "kernels.hh"
#include <type_traits>

namespace awmk {

namespace {

constexpr int NB = 81;
typedef const int __attribute__ ((address_space (4))) *const_int_ptr;
typedef const unsigned __attribute__ ((address_space (4))) *const_uint_ptr;
typedef int int4v __attribute__ ((ext_vector_type (4)));

constexpr int QUAD_TILE = 252;          // candidates a wave delivers (see above)

/* acc + (value the lane above holds): the compiler folds the DPP move into v_add_f32_dpp; rounding == __fadd_rn */
__device__ __forceinline__ float
add_from_upper_lane (float acc, float v)
{
  const int up = __builtin_amdgcn_update_dpp (0, __float_as_int (v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
  return __fadd_rn (acc, __int_as_float (up));
}

/* terms of one sync frame: J = (frame mod 4) selects which elements of the aligned quad are the lane's own */
template<int J, int N> __device__ __forceinline__ void
add_terms (float (&acc)[4], const float4 (&v)[N])
{
#pragma unroll
  for (int i = 0; i < N; i++)
    {
      const float e[4] = { v[i].x, v[i].y, v[i].z, v[i].w };
#pragma unroll
      for (int c = 0; c < 4; c++)
        {
          if (c + J < 4)
            acc[c] = __fadd_rn (acc[c], e[c + J]);
          else
            acc[c] = add_from_upper_lane (acc[c], e[c + J - 4]);
        }
    }
}

constexpr int NCHAIN = 12;
constexpr int CH = 64;                  // frames per chunk
constexpr int SLOTS = 7;
constexpr int RINGF = SLOTS * CH;
constexpr int BANDS_PAD = 84;           // 21 DMA pieces of 4 bands
constexpr int SLOT_FLOATS = BANDS_PAD * CH;
constexpr int PIECES = BANDS_PAD / 4;
constexpr int INFLIGHT = 2;             // chunks the loader has in flight
static_assert (RINGF >= 256 + CH + 3, "the slowest chain's 256 + 3 frames fit beside the chunk being replaced");
static_assert (PIECES * INFLIGHT < 64, "vmcnt counts 63 outstanding operations");

}  // namespace

/* HAVE (CLIP mode): the frames of the plane that were transformed form ONE run [F0, F1) (the rest is the leading / trailing
 * silence of a padded clip, syncfinder.cc:155-169, 578-590; their matrix entries are +0).  A sync frame none of the tile's
 * candidates has inside the run is skipped by its chain, chunks no live sync frame can touch are not loaded; inside a live
 * sync frame the candidates outside the run add +0, which leaves their sums as they are.  The frame counts that weight the
 * bits (frame_bit_count, syncfinder.cc:147-152) are counted per candidate at the end. */
template<bool HAVE> __global__ void __launch_bounds__ (832)
sync_scan_stream_kernel (SyncScanArgs a, int total_frames)
{
  __shared__ __attribute__ ((aligned (16))) float s_win[SLOTS * SLOT_FLOATS];
  __shared__ __attribute__ ((aligned (16))) int s_progress[16];   // per chain: frame of its next sync frame (first frame it still needs); [12..15] = "done"
  __shared__ int s_loaded;                                 // frames [.., s_loaded) have landed in the ring
  __shared__ int s_run[2];                                 // HAVE: [F0, F1)
  const int lane = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane (threadIdx.y);
  const int tid = threadIdx.y * 64 + threadIdx.x;
  const long long plane = blockIdx.y;
  // XCD-aware tile order: workgroup b runs on XCD b % 8, every XCD gets one contiguous range of tiles so that the
  // overlap of neighbouring tiles' frame ranges (90 %) is served by that XCD's L2
  const long long n_tiles = (a.n_lanes + QUAD_TILE - 1) / QUAD_TILE;
  const long long per_xcd = (n_tiles + 7) / 8;
  const long long tile = (long long) (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if ((blockIdx.x >> 3) >= per_xcd || tile >= n_tiles)
    return;                                               // uniform for the workgroup
  const long long sf0 = tile * QUAD_TILE;                 // multiple of 4
  const float *db = a.db + plane * a.plane_stride;
  const long long ld = a.band_stride;                     // multiple of 64
  const int R = a.table.rows_per_bit;

  if (tid < 16)
    s_progress[tid] = tid < NCHAIN ? 0 : 0x7fffffff;
  if (tid == 16)
    s_loaded = 0;
  if (tid == 17)
    {
      s_run[0] = 0x7fffffff;
      s_run[1] = 0;
    }
  __syncthreads();
  int run0 = 0, run1 = 0x7fffffff;                         // the run, relative to the tile's first frame
  if (HAVE)
    {
      const char *have = a.have + plane * a.have_plane_stride;
      const int n_have = int (a.n_lanes) + total_frames;   // frames any candidate of the plane can touch
      int lo = 0x7fffffff, hi = 0;
      for (int f = tid; f < n_have; f += 832)
        if (have[f])
          {
            lo = min (lo, f);
            hi = max (hi, f + 1);
          }
      for (int o = 32; o > 0; o >>= 1)
        {
          lo = min (lo, __shfl_xor (lo, o));
          hi = max (hi, __shfl_xor (hi, o));
        }
      if (lane == 0 && hi > 0)
        {
          atomicMin (&s_run[0], lo);
          atomicMax (&s_run[1], hi);
        }
      __syncthreads();
      run0 = __builtin_amdgcn_readfirstlane (s_run[0]) - int (sf0);
      run1 = __builtin_amdgcn_readfirstlane (s_run[1]) - int (sf0);
      if (__builtin_amdgcn_readfirstlane (s_run[1]) == 0)
        run0 = run1 = 0;                                    // nothing transformed at all: no sync frame is live
    }
  // a sync frame at (tile relative) frame fr is live if one of the tile's 256 candidates has it inside the run
  auto live = [&] (int fr) { return !HAVE || (fr + 255 >= run0 && fr < run1); };

  float acc[4] = { 0.f, 0.f, 0.f, 0.f };
  if (wv == NCHAIN)
    {
      /* ---- loader ---- */
      const int n_chunks = (total_frames + 256 + CH - 1) / CH;            // every chain's last sync frame + 256 lies below
      const int lb = lane >> 4, lq = lane & 15;                           // lane -> (band within the piece, 4 frames)
      auto issue = [&] (int c) {
        const long long f0 = sf0 + (long long) c * CH;
        // frames past the matrix are only seen by candidates that are dropped: read something that exists instead
        const long long f = f0 + 4 * lq + 4 <= ld ? f0 + 4 * lq : 0;
        float *dst = s_win + (c % SLOTS) * SLOT_FLOATS;
#pragma unroll
        for (int u = 0; u < PIECES; u++)
          {
            const int band = min (4 * u + lb, NB - 1);                    // the pad bands re-read band 80
            __builtin_amdgcn_global_load_lds (db + band * ld + f, dst + u * 4 * CH, 16, 0, 0);
          }
      };
      // The loader's own LDS accesses are written in assembly: hipcc orders every LDS access it knows of behind ALL
      // outstanding LDS-DMA of the wave (s_waitcnt vmcnt(0)), which would leave no chunk in flight.
      const unsigned progress_addr = (unsigned) (uintptr_t) (__attribute__ ((address_space (3))) int *) s_progress;
      const unsigned loaded_addr = (unsigned) (uintptr_t) (__attribute__ ((address_space (3))) int *) &s_loaded;
      int published = 0;
      auto publish = [&] (int c) {
        if ((c + 1) * CH <= published)
          return;
        published = (c + 1) * CH;
        asm volatile ("ds_write_b32 %0, %1" :: "v" (loaded_addr), "v" (published) : "memory");
      };
      auto wait_room = [&] (int c) {
        // the slot of chunk c still holds chunk c - SLOTS: wait until no chain needs its frames (a chain at frame f reads from f - 3 on)
        const int need = (c + 1 - SLOTS) * CH + 3;
        for (;;)
          {
            int4v p0, p1, p2;
            asm volatile ("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %3 offset:32\n\ts_waitcnt lgkmcnt(0)"
                          : "=&v" (p0), "=&v" (p1), "=&v" (p2) : "v" (progress_addr) : "memory");
            const int m = min (min (min (p0.x, p0.y), min (p0.z, p0.w)), min (min (min (p1.x, p1.y), min (p1.z, p1.w)), min (min (p2.x, p2.y), min (p2.z, p2.w))));
            if (__builtin_amdgcn_readfirstlane (m) >= need)
              break;
            // a chain that is about to need the chunks in flight must not wait for THIS wave: publish them before sleeping
            asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
            publish (c - 1);
            __builtin_amdgcn_s_sleep (1);
          }
      };
      // HAVE: live sync frames read the frames [run0 - 258, run1 + 255] of the tile at most: only these chunks are loaded
      int c_lo = 0, c_hi = n_chunks;
      if (HAVE)
        {
          c_lo = min (n_chunks, max (0, (run0 - 258) >> 6));
          c_hi = run1 > run0 ? min (n_chunks, max (c_lo, ((run1 + 255) >> 6) + 1)) : c_lo;
          if (c_lo > 0)
            publish (c_lo - 1);                            // nothing below is ever waited for by a live sync frame
        }
      for (int c = c_lo; c < c_hi; c++)
        {
          if (c >= c_lo + INFLIGHT)
            {
              asm volatile ("s_waitcnt vmcnt(%0)" :: "n" (PIECES * (INFLIGHT - 1)) : "memory");   // vmcnt counts the pieces in issue order:
              publish (c - INFLIGHT);                                                             // all but the youngest chunk(s) have landed
            }
          if (c >= c_lo + SLOTS)
            wait_room (c);
          issue (c);
        }
      asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
      publish (HAVE ? 0xffffff : n_chunks - 1);       // (HAVE: whatever a chain may still ask for)
    }
  else
    {
      /* ---- one chain ---- */
      // The row descriptors come through the VECTOR memory path (every lane reads the same 32 bytes, the values go to scalar registers
      // with v_readfirstlane when the row starts): a scalar load shares its counter with the LDS gathers and returns out of order, so
      // hipcc waits for it right after issuing it (s_waitcnt lgkmcnt(0)) -- round 2's "requested a row ahead" cost the full load latency
      // per row, 40 % of the kernel.  vmcnt belongs to the descriptors alone in a chain wave.
      int vzero = 0;
      asm volatile ("" : "+v" (vzero));                    // (a uniform address would be turned back into a scalar load)
      const unsigned *tab = a.table.chains + (size_t) (blockIdx.y / a.table.planes_per_slice) * a.table.chains_slice_stride
                          + (size_t) wv * R * 8 + vzero;
      int loaded = 0;
      uint4 rowdesc[2][2];                                 // 30 band bytes + u16 frame of the next row
      auto load_row = [&] (uint4 (&w)[2], int r) {
        const uint4 *tr = reinterpret_cast<const uint4 *> (tab + r * 8);
        w[0] = tr[0];
        w[1] = tr[1];
      };
      auto to_scalar = [] (const uint4 (&v)[2], unsigned (&w)[8]) {
        const unsigned e[8] = { v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w };
#pragma unroll
        for (int i = 0; i < 8; i++)
          w[i] = __builtin_amdgcn_readfirstlane (e[i]);
      };
      auto row = [&] (const unsigned (&w)[8], int fr) {
        if (!live (fr))
          return;
        while (loaded < fr + 256)
          {
            loaded = __hip_atomic_load (&s_loaded, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (loaded < fr + 256)
              __builtin_amdgcn_s_sleep (1);
          }
        asm volatile ("" ::: "memory");
        const int j = fr & 3;
        int phys = (fr - j) % RINGF + 4 * lane;
        phys -= phys >= RINGF ? RINGF : 0;
        const char *base = reinterpret_cast<const char *> (s_win + (phys >> 6) * SLOT_FLOATS + (phys & 63));
        // 30 gathers + 120 additions in the reference's order, software-pipelined: the gathers of the next batch of bands are in
        // flight while the additions of this one issue (two buffers of 8 quads; round 2: three batches of 10, each one waiting for
        // its own gathers first -- the switch over the alignment stood between a batch's additions and the next batch's gathers)
        auto gather = [&] (float4 (&v)[8], int t0, int n) {
#pragma unroll
          for (int i = 0; i < 8; i++)
            if (i < n)
              {
                const int t = t0 + i;
                const unsigned band = (w[t >> 2] >> (8 * (t & 3))) & 0xff;
                v[i] = *reinterpret_cast<const float4 *> (base + (band << 8));        // a band of a chunk = 64 floats
              }
        };
        auto body = [&] (auto jc) {
          constexpr int J = decltype (jc)::value;
          float4 va[8], vb[8];
          gather (va, 0, 8);
          gather (vb, 8, 8);
          add_terms<J, 8> (acc, va);
          gather (va, 16, 8);
          add_terms<J, 8> (acc, vb);
          gather (vb, 24, 6);
          add_terms<J, 8> (acc, va);
          float4 vl[6];
#pragma unroll
          for (int i = 0; i < 6; i++)
            vl[i] = vb[i];
          add_terms<J, 6> (acc, vl);
        };
        switch (j)
          {
          case 0:  body (std::integral_constant<int, 0>()); break;
          case 1:  body (std::integral_constant<int, 1>()); break;
          case 2:  body (std::integral_constant<int, 2>()); break;
          default: body (std::integral_constant<int, 3>()); break;
          }
      };
      auto done_with = [&] (int next_frame) {
        // the gathers of this row have returned (their values were consumed): its frames may be replaced
        asm volatile ("" ::: "memory");
        if (lane == 0)
          __hip_atomic_store (&s_progress[wv], next_frame, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      };
      auto next_frame = [] (const unsigned (&w)[8]) { const int f = int (w[7] >> 16); return f == 0xffff ? 0x7fffffff : f; };
      // rows [r_lo, r_hi) of this chain: all of them, or (HAVE) the live ones -- the frames of a bit's rows ascend.  (Walking the
      // silent rows one by one costs a scalar load latency each with nothing to hide it behind: 3 x the time of the live ones.)
      const int fstride = a.table.row_frames ? 1 : 64;
      const_int_ptr frames = a.table.row_frames
                           ? ((const_int_ptr) a.table.row_frames) + ((size_t) (blockIdx.y / a.table.planes_per_slice) * 6 + (wv >> 1)) * R
                           : ((const_int_ptr) a.table.packed) + (size_t) (wv >> 1) * R * 64 + 60;
      int r_lo = 0, r_hi = R;
      if (HAVE)
        {
          auto rows_below = [&] (int x) {
            int lo = 0, hi = R;
            while (lo < hi)
              {
                const int mid = (lo + hi) >> 1;
                if (frames[(size_t) mid * fstride] < x)
                  lo = mid + 1;
                else
                  hi = mid;
              }
            return lo;
          };
          r_lo = rows_below (run0 - 255);
          r_hi = run1 > run0 ? rows_below (run1) : r_lo;
        }
      int fr = r_lo < r_hi ? frames[(size_t) r_lo * fstride] : 0x7fffffff;
      if (r_lo < r_hi)
        {
          done_with (fr);
          load_row (rowdesc[0], r_lo);
        }
      for (int r = r_lo; r < r_hi; r += 2)
        {
          unsigned w[8];
          to_scalar (rowdesc[0], w);
          if (r + 1 < r_hi)
            load_row (rowdesc[1], r + 1);
          row (w, fr);
          fr = next_frame (w);
          done_with (fr);
          if (r + 1 < r_hi)
            {
              to_scalar (rowdesc[1], w);
              if (r + 2 < r_hi)
                load_row (rowdesc[0], r + 2);
              row (w, fr);
              fr = next_frame (w);
              done_with (fr);
            }
        }
      done_with (0x7fffffff);
    }
  __syncthreads();                                         // ring is dead: chain sums -> s_sum[chain][candidate of the tile]

  float (*s_sum)[256] = reinterpret_cast<float (*)[256]> (s_win);
  int *s_frames = reinterpret_cast<int *> (s_win + NCHAIN * 256);      // HAVE: frame of every row, [bit][R]
  if (wv < NCHAIN)
    *reinterpret_cast<float4 *> (&s_sum[wv][4 * lane]) = make_float4 (acc[0], acc[1], acc[2], acc[3]);
  if (HAVE)
    for (int i = tid; i < 6 * R; i += 832)
      s_frames[i] = a.table.row_frames ? a.table.row_frames[(size_t) (blockIdx.y / a.table.planes_per_slice) * 6 * R + i] : a.table.packed[(size_t) i * 64 + 60];
  __syncthreads();
  if (tid < QUAD_TILE && sf0 + tid < a.n_lanes)
    {
      double q = 0;
      int total = 0;
      for (int b = 0; b < 6; b++)
        {
          int n_b = R;                                    // frame_bit_count: sync frames of the bit inside the run
          if (HAVE)
            {
              // the frames of a bit's rows ascend: rows with frame < x, twice
              const int *frames = s_frames + b * R;
              auto rows_below = [&] (int x) {
                int lo = 0, hi = R;
                while (lo < hi)
                  {
                    const int mid = (lo + hi) >> 1;
                    if (frames[mid] < x)
                      lo = mid + 1;
                    else
                      hi = mid;
                  }
                return lo;
              };
              n_b = rows_below (run1 - tid) - rows_below (run0 - tid);
            }
          const float um = s_sum[2 * b][tid], dm = s_sum[2 * b + 1][tid];
          // SyncFinder::bit_quality (reference syncfinder.cc:94-114): float division and subtraction
          float raw;
          if (um == 0 || dm == 0)
            raw = 0;
          else if (um < dm)
            raw = __fsub_rn (1.f, __fdiv_rn (um, dm));
          else
            raw = __fsub_rn (__fdiv_rn (dm, um), 1.f);
          const double rb = (b & 1) ? double (raw) : -double (raw);
          q += rb * n_b;
          total += n_b;
        }
      if (total)
        q /= total;
      q = q / a.min_delta / 2.9;
      a.quality[plane * a.q_stride + sf0 + tid] = q;
    }
}

void
pack_scan_chains (const int *packed, int rows_per_bit, unsigned *out)
{
  const int R = rows_per_bit;
  for (int chain = 0; chain < 12; chain++)
    for (int r = 0; r < R; r++)
      {
        const int *row = packed + ((size_t) (chain >> 1) * R + r) * 64;
        unsigned char bytes[32];
        for (int i = 0; i < 30; i++)
          bytes[i] = (unsigned char) row[30 * (chain & 1) + i];
        const unsigned next = r + 1 < R ? unsigned (row[61]) : 0xffffu;
        bytes[30] = next & 0xff;
        bytes[31] = next >> 8;
        for (int i = 0; i < 8; i++)
          out[((size_t) chain * R + r) * 8 + i] = bytes[4 * i] | bytes[4 * i + 1] << 8 | bytes[4 * i + 2] << 16 | unsigned (bytes[4 * i + 3]) << 24;
      }
}

hipError_t
launch_sync_scan_window (hipStream_t st, const SyncScanArgs& a, int total_frames)
{
  if (a.n_lanes <= 0 || a.n_planes <= 0)
    return hipSuccess;
  if (a.row_stride != 1 || a.n_planes > 65535 || (a.band_stride & 63) || a.lane_count)
    return hipErrorInvalidValue;
  if ((a.have && !a.have_is_run) || !a.table.chains || total_frames >= 0xffff || a.n_lanes + total_frames > 0x1000000 || a.table.rows_per_bit > 4096)
    {
      if (a.table.chains_slice_stride)
        return hipErrorInvalidValue;                      // (per-slice key tables exist for this kernel only)
      return launch_sync_scan (st, a);                    // arbitrary skipped frames: the generic kernel handles any `have`
    }
  const long long px = ((a.n_lanes + QUAD_TILE - 1) / QUAD_TILE + 7) / 8;
  const dim3 grid ((unsigned) (px * 8), (unsigned) a.n_planes), block (64, NCHAIN + 1);
  if (a.have)
    hipLaunchKernelGGL (sync_scan_stream_kernel<true>, grid, block, 0, st, a, total_frames);
  else
    hipLaunchKernelGGL (sync_scan_stream_kernel<false>, grid, block, 0, st, a, total_frames);
  return hipGetLastError();
}

}  // namespace awmk
