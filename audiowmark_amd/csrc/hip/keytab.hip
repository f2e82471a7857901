// keytab.hip -- K16: the frame_mod table of `add` built ON THE DEVICE, one workgroup per key; K16g: the same draws and shuffles turned
// into the tables `get` needs for a clip with a key of its own (clip_key_tables_tail below).
//
// Replaces, for batches with one key per clip (awm_add_watermark_batch_keys_d), the host's build_frame_mod_table
// (host/wmcommon.cc; reference wmadd.cc:86-162 "init_frame_mod_vec", wmcommon.cc:143-202 UpDownGen / BitPosGen / gen_mix_entries,
// random.cc:97-161 the AES-128-CTR generator, random.hh:102-113 the shuffle).  A table costs ~1 ms of one host core and a process on
// the box may use 16 of them: 1024 keys were 64 ms of wall time in front of 25 ms of device work.  Per key the work is
//   235 000 pseudo random draws = 118 000 AES-128 blocks in counter mode          -- independent: all lanes of the workgroup
//   2226 + 1716 shuffles of the 81 bands (one per sync / data frame and its seed)  -- independent: one lane per frame
//   three Fisher-Yates shuffles over the key's own streams (2226 frame positions, 858 coded bits, 51 480 mix entries)
//     j = i + draw_i mod (n - i): the draws do not depend on the array, so all TARGETS j_i are computed in parallel; only the
//     swaps are sequential -- but independent of each other far more often than not: a chunk of 64 steps whose targets all lie
//     beyond the chunk and differ goes through in ONE LDS round trip, every lane its own swap (80 % of the chunks; the test is
//     exact), else a quarter of 16 the same way, else four steps per round trip after a scalar test, else one by one.  The
//     permutation lives in LDS (51 480 x u16 = 103 KB: one key per compute unit): one swap per round trip took 3.2 ms, the chunks with
//     their tests inside the sequential wave 1.1 ms; 1.16 ms per launch of 256 keys now (round 4: 1.95; round 5: 512 lanes per key, the
//     T-table once per LDS bank, the independence tests by all waves ahead of the swaps, four chunks of targets read ahead, the
//     table's entries eight at a time with their loads first); the tables of the next 256 are built while the first 256 clips are watermarked
//   the table itself: 2 x 2226 x 81 bytes, every (frame, band) written at most once -- all lanes.
// Bit-identical to the host's tables (tests/test_gpu_parity.py::test_key_tables_on_the_device).
#include "kernels.hh"

namespace awmk {

namespace {

constexpr int KT_NB = 81, KT_MIN_BAND = 20, KT_BPF = 30;
constexpr int KT_SYNC = 510, KT_DATA = 1716, KT_BLOCK = KT_SYNC + KT_DATA;        // frames
constexpr int KT_SYNC_FPB = 85, KT_CODED = 858, KT_FPB = 2;
constexpr int KT_MIX = KT_DATA * KT_BPF;                                           // 51 480
constexpr int KT_WG = 512;

struct Aes
{
  // The T-table (2 s, s, s, 3 s per S-box entry) once PER LDS BANK: te[32 x + bank], a lane reads the copy of its own bank (lane mod 32).
  // The 64 lanes of a lookup name 64 unrelated entries: from ONE table they collide in the 32 banks five or six deep (measured: 2.1 us
  // per block and wave, the 160 lookups of a block bound by the LDS); with the copies no two lanes of a half wave share a bank.  The
  // S-box of the last round is the table's second byte.
  const unsigned int *te;        // LDS: the lane's column of the 256 x 32 words
  unsigned int rk[44];           // big-endian round key words

  __device__ __forceinline__ unsigned int te0 (unsigned int x) const { return te[x << 5]; }
  __device__ __forceinline__ unsigned int sbox (unsigned int x) const { return (te[x << 5] >> 8) & 0xff; }
  __device__ __forceinline__ static unsigned int rotr (unsigned int v, int n) { return (v >> n) | (v << (32 - n)); }

  /* AES-128 of the block (s0 .. s3, big-endian words) */
  __device__ __forceinline__ void
  encrypt (unsigned int& s0, unsigned int& s1, unsigned int& s2, unsigned int& s3) const
  {
    s0 ^= rk[0]; s1 ^= rk[1]; s2 ^= rk[2]; s3 ^= rk[3];
#pragma unroll
    for (int r = 1; r < 10; r++)                          // (unrolled: the round keys stay in scalar registers)
      {
        const unsigned int t0 = te0 (s0 >> 24) ^ rotr (te0 ((s1 >> 16) & 0xff), 8) ^ rotr (te0 ((s2 >> 8) & 0xff), 16) ^ rotr (te0 (s3 & 0xff), 24) ^ rk[4 * r];
        const unsigned int t1 = te0 (s1 >> 24) ^ rotr (te0 ((s2 >> 16) & 0xff), 8) ^ rotr (te0 ((s3 >> 8) & 0xff), 16) ^ rotr (te0 (s0 & 0xff), 24) ^ rk[4 * r + 1];
        const unsigned int t2 = te0 (s2 >> 24) ^ rotr (te0 ((s3 >> 16) & 0xff), 8) ^ rotr (te0 ((s0 >> 8) & 0xff), 16) ^ rotr (te0 (s1 & 0xff), 24) ^ rk[4 * r + 2];
        const unsigned int t3 = te0 (s3 >> 24) ^ rotr (te0 ((s0 >> 16) & 0xff), 8) ^ rotr (te0 ((s1 >> 8) & 0xff), 16) ^ rotr (te0 (s2 & 0xff), 24) ^ rk[4 * r + 3];
        s0 = t0; s1 = t1; s2 = t2; s3 = t3;
      }
    auto sub = [&] (unsigned int a, unsigned int b, unsigned int c, unsigned int d) {
      return (sbox (a >> 24) << 24) | (sbox ((b >> 16) & 0xff) << 16) | (sbox ((c >> 8) & 0xff) << 8) | sbox (d & 0xff);
    };
    const unsigned int t0 = sub (s0, s1, s2, s3) ^ rk[40], t1 = sub (s1, s2, s3, s0) ^ rk[41], t2 = sub (s2, s3, s0, s1) ^ rk[42], t3 = sub (s3, s0, s1, s2) ^ rk[43];
    s0 = t0; s1 = t1; s2 = t2; s3 = t3;
  }
};

/* the generator of one (seed, stream): counter block = AES (seed as big-endian u64 || stream || 7 zero bytes); key stream block n =
 * AES (counter + n), the counter a 128 bit big-endian integer; draws 2 n, 2 n + 1 = the block's two big-endian u64 (random.cc:97-161) */
struct Stream
{
  unsigned int c0, c1, c2, c3;
  __device__ __forceinline__ void
  seed (const Aes& aes, unsigned long long seed, unsigned int stream)
  {
    c0 = (unsigned int) (seed >> 32); c1 = (unsigned int) seed; c2 = stream << 24; c3 = 0;
    aes.encrypt (c0, c1, c2, c3);
  }
  __device__ __forceinline__ void
  block (const Aes& aes, unsigned int n, unsigned long long& d0, unsigned long long& d1) const
  {
    unsigned int s3 = c3 + n;
    unsigned int carry = s3 < n;
    unsigned int s2 = c2 + carry;
    carry = carry && s2 == 0;
    unsigned int s1 = c1 + carry;
    carry = carry && s1 == 0;
    unsigned int s0 = c0 + carry;
    aes.encrypt (s0, s1, s2, s3);
    d0 = ((unsigned long long) s0 << 32) | s1;
    d1 = ((unsigned long long) s2 << 32) | s3;
  }
};

/* x mod d for a 64 bit x and d < 65536, in 32 bit steps */
__device__ __forceinline__ unsigned int
mod_small (unsigned long long x, unsigned int d)
{
  unsigned int r = (unsigned int) (x >> 32) % d;
  r = ((r << 16) | ((unsigned int) (x >> 16) & 0xffff)) % d;
  r = ((r << 16) | ((unsigned int) x & 0xffff)) % d;
  return r;
}

/* the sequential half of a Fisher-Yates shuffle: swap (v[i], v[target[i]]) for i = 0 .. n - 1, by ONE lane */
template<class T, class TGT> __device__ __forceinline__ void
apply_swaps (T *v, const TGT *target, int n)
{
  for (int i = 0; i < n; i++)
    {
      const int j = target[i];
      const T a = v[i], b = v[j];
      v[i] = b;
      v[j] = a;
    }
}

}  // namespace

/* scratch per key (global memory): [KT_MIX] u16 mix targets, [KT_BLOCK][60] u8 up / down bands of every frame's seed (sync frames
 * first), rounded up */
constexpr size_t KT_SCRATCH_BYTES = ((size_t (KT_MIX) * 2 + size_t (KT_BLOCK) * 60 + 255) / 256) * 256;
size_t key_table_scratch_bytes() { return KT_SCRATCH_BYTES; }
size_t key_table_bytes() { return size_t (2) * KT_BLOCK * KT_NB; }

static_assert (CLIP_KEY_ROWS == 2 * KT_SYNC_FPB && CLIP_KEY_WANT == 2 * KT_SYNC && CLIP_KEY_MIX == KT_MIX && CLIP_KEY_CODED == KT_CODED, "clip key table shapes");

/* K16g's own part, after the shuffles: s_perm = the mix permutation, s_pos = the frame positions, s_order = the bit order (LDS),
 * updown = [frame: 510 sync, 1716 data][30 up, 30 down] bands as drawn (global scratch, written by this workgroup) */
__device__ __forceinline__ void
clip_key_tables_tail (const ClipKeyTableOut& o, long long key, int tid, unsigned short *s_perm, const unsigned short *s_pos, const unsigned short *s_order,
                      const unsigned char *updown)
{
  constexpr int R = CLIP_KEY_ROWS, NW = CLIP_KEY_WANT;
  // ---- mix entries in shuffled order (wmcommon.cc build_mix_table) and the inverse bit order
  // (eight entries per lane at a time, their loads ahead of the stores: see the table fill of K16)
  constexpr int UB = 8;
  for (int p0 = tid; p0 < KT_MIX; p0 += KT_WG * UB)
    {
      int pos[UB], up[UB], down[UB];
#pragma unroll
      for (int k = 0; k < UB; k++)
        {
          const int e = s_perm[min (p0 + k * KT_WG, KT_MIX - 1)];
          const int f = e / KT_BPF, i = e % KT_BPF;
          const unsigned char *ud = updown + size_t (KT_SYNC + f) * 60;
          pos[k] = s_pos[KT_SYNC + f];
          up[k] = ud[i];
          down[k] = ud[30 + i];
        }
#pragma unroll
      for (int k = 0; k < UB; k++)
        if (p0 + k * KT_WG < KT_MIX)
          {
            const int p = p0 + k * KT_WG;
            o.mix_frame[key * KT_MIX + p] = short (pos[k]);
            o.mix_up[key * KT_MIX + p] = (unsigned char) up[k];
            o.mix_down[key * KT_MIX + p] = (unsigned char) down[k];
          }
    }
  for (int i = tid; i < KT_CODED; i += KT_WG)
    o.inv_order[key * KT_CODED + s_order[i]] = i;
  __syncthreads();                                          // (the permutation is done with: its memory holds the lists below)
  // ---- the 510 sync frames' bands, each list ascending and relative to the first band (wmcommon.cc build_sync_table; the reference
  // sorts them, syncfinder.cc:52-53): the 30 bands of a list are distinct members of 0 .. 80 -- mark and scan
  unsigned char (*s_list)[60] = reinterpret_cast<unsigned char (*)[60]> (s_perm);                 // 30 600 bytes
  unsigned short *s_rf = s_perm + 16384;                                                          // [6][170] frame of a row (from byte 32 768)
  unsigned short *s_item = s_rf + 6 * R;                                                          // [6][170] (sync frame, block) of a row
  for (int sf = tid; sf < KT_SYNC; sf += KT_WG)
    for (int half = 0; half < 2; half++)
      {
        const unsigned char *ud = updown + size_t (sf) * 60 + 30 * half;
        unsigned int m[3] = { 0, 0, 0 };
        for (int i = 0; i < 30; i++)
          {
            const int b = ud[i] - KT_MIN_BAND;
            m[b >> 5] |= 1u << (b & 31);
          }
        int n = 0;
        for (int w = 0; w < 3; w++)
          for (unsigned int bits = m[w]; bits; bits &= bits - 1)
            s_list[sf][30 * half + n++] = (unsigned char) (32 * w + __builtin_ctz (bits));
      }
  __syncthreads();
  // ---- a row = (sync frame, block 0 | 1 of the long block); within its bit the rows are ordered by frame, the want list orders all
  // 1020: block 1 lies behind block 0, so both ranks come from comparing the 510 positions.  Block 1 carries the inverted sequence:
  // its up list is the frame's down list.
  for (int item = tid; item < NW; item += KT_WG)
    {
      const int block = item >= KT_SYNC, sf = item - KT_SYNC * block, bit = sf / KT_SYNC_FPB;
      const int my = s_pos[sf];
      int r = 0, w = 0;
      for (int j = 0; j < KT_SYNC_FPB; j++)
        r += s_pos[bit * KT_SYNC_FPB + j] < my;
      for (int j = 0; j < KT_SYNC; j++)
        w += s_pos[j] < my;
      r += block * KT_SYNC_FPB;
      w += block * KT_SYNC;
      const int frame = my + block * KT_BLOCK, src = bit * R + r;
      s_rf[src] = (unsigned short) frame;
      s_item[src] = (unsigned short) item;
      o.row_frames[key * NW + src] = frame;
      o.want[key * NW + w] = frame;
      o.perm[key * NW + w] = src;
      unsigned char *prow = o.pos + (key * NW + w) * KT_NB;
      for (int b = 0; b < KT_NB; b++)
        prow[b] = 255;
      const unsigned char *up = s_list[sf] + 30 * block, *down = s_list[sf] + 30 * (1 - block);
      for (int i = 0; i < 30; i++)
        {
          prow[up[i]] = (unsigned char) i;
          prow[down[i]] = (unsigned char) (30 + i);
        }
    }
  __syncthreads();
  // ---- K5w's chains (scan.hip pack_scan_chains): per (bit, up | down) the rows by frame, 30 band bytes + the NEXT row's frame
  for (int src = tid; src < 6 * R; src += KT_WG)
    {
      const int bit = src / R, r = src - bit * R;
      const int item = s_item[src], block = item >= KT_SYNC, sf = item - KT_SYNC * block;
      const unsigned int next = r + 1 < R ? s_rf[src + 1] : 0xffffu;
      for (int ud = 0; ud < 2; ud++)
        {
          const unsigned char *list = s_list[sf] + 30 * (ud ^ block);
          unsigned int *out = o.chains + ((key * 12 + 2 * bit + ud) * R + r) * 8;
          for (int i = 0; i < 7; i++)
            out[i] = list[4 * i] | list[4 * i + 1] << 8 | list[4 * i + 2] << 16 | (unsigned int) list[4 * i + 3] << 24;
          out[7] = list[28] | list[29] << 8 | (next & 0xff) << 16 | (next >> 8) << 24;
        }
    }
}

template<bool GET> __device__ __forceinline__ void
key_tables_body (const KeyTableArgs& a, const ClipKeyTableOut& o)
{
  __shared__ unsigned short s_perm[KT_MIX];                // the mix shuffle's array: entry numbers
  __shared__ unsigned short s_pos[KT_BLOCK], s_pos_t[KT_BLOCK];      // frame positions and their swap targets
  __shared__ unsigned short s_order[KT_CODED], s_order_t[KT_CODED];  // bit order and its swap targets
  // a lane's 81 bands while it shuffles them: in s_perm's memory, which is not needed before the swaps -- 117 KB in all, so that a
  // workgroup of the fused add kernel (39 KB) fits on the compute unit beside this one: the clips of the previous group of keys are
  // watermarked WHILE this group's tables are built (with 138 KB the two kernels took turns: 49.8 instead of 31 ms for 1024 clips)
  unsigned char (*s_bands)[84] = reinterpret_cast<unsigned char (*)[84]> (s_perm);
  static_assert (sizeof (unsigned char[KT_WG][84]) <= sizeof (unsigned short[KT_MIX]), "the band arrays fit into the permutation's memory");
  const int tid = threadIdx.x;
  const long long key = blockIdx.x;
  // (the 32 copies of the T-table live in the permutation's memory as well, behind the band arrays: the generator is done when the swaps start)
  unsigned int *s_te = reinterpret_cast<unsigned int *> (s_perm) + 11264;                 // 32 KB from byte 45 056
  static_assert (sizeof (unsigned char[KT_WG][84]) <= 45056 && 45056 + 32768 <= sizeof (unsigned short[KT_MIX]), "band arrays and T-tables fit into the permutation's memory");
  for (int i = tid; i < 256 * 32; i += KT_WG)
    {
      const unsigned int s = a.sbox[i >> 5];
      const unsigned int s2 = ((s << 1) ^ ((s & 0x80) ? 0x1b : 0)) & 0xff;
      s_te[i] = (s2 << 24) | (s << 16) | (s << 8) | (s2 ^ s);
    }
  Aes aes;
  aes.te = s_te + (tid & 31);
  {
    const unsigned char *rk = a.round_keys + key * 176;
#pragma unroll
    for (int i = 0; i < 44; i++)
      aes.rk[i] = ((unsigned int) rk[4 * i] << 24) | ((unsigned int) rk[4 * i + 1] << 16) | ((unsigned int) rk[4 * i + 2] << 8) | rk[4 * i + 3];
  }
  unsigned char *scratch = a.scratch + (key % a.scratch_slots) * KT_SCRATCH_BYTES;
  unsigned short *mix_t = reinterpret_cast<unsigned short *> (scratch);
  unsigned char *updown = scratch + size_t (KT_MIX) * 2;                     // [frame: 510 sync, then 1716 data][60]
  for (int i = tid; i < KT_BLOCK; i += KT_WG)
    s_pos[i] = (unsigned short) i;
  for (int i = tid; i < KT_CODED; i += KT_WG)
    s_order[i] = (unsigned short) i;
  __syncthreads();

  // ---- the swap targets of the three key-wide shuffles (streams frame_position = 6, bit_order = 5, mix = 4; seed 0)
  {
    Stream st;
    st.seed (aes, 0, 6);
    for (int b = tid; b < (KT_BLOCK + 1) / 2; b += KT_WG)
      {
        unsigned long long d0, d1;
        st.block (aes, b, d0, d1);
        const int i0 = 2 * b, i1 = 2 * b + 1;
        s_pos_t[i0] = (unsigned short) (i0 + mod_small (d0, KT_BLOCK - i0));
        if (i1 < KT_BLOCK)
          s_pos_t[i1] = (unsigned short) (i1 + mod_small (d1, KT_BLOCK - i1));
      }
    st.seed (aes, 0, 5);
    for (int b = tid; b < (KT_CODED + 1) / 2; b += KT_WG)
      {
        unsigned long long d0, d1;
        st.block (aes, b, d0, d1);
        const int i0 = 2 * b, i1 = 2 * b + 1;
        s_order_t[i0] = (unsigned short) (i0 + mod_small (d0, KT_CODED - i0));
        if (i1 < KT_CODED)
          s_order_t[i1] = (unsigned short) (i1 + mod_small (d1, KT_CODED - i1));
      }
    st.seed (aes, 0, 4);
    for (int b = tid; b < KT_MIX / 2; b += KT_WG)
      {
        unsigned long long d0, d1;
        st.block (aes, b, d0, d1);
        const int i0 = 2 * b, i1 = 2 * b + 1;
        mix_t[i0] = (unsigned short) (i0 + mod_small (d0, KT_MIX - i0));
        mix_t[i1] = (unsigned short) (i1 + mod_small (d1, KT_MIX - i1));
      }
  }
  // ---- up / down bands of every frame's seed: UpDownGen::get (f) = the first 60 of the 81 bands shuffled with the stream of
  // (seed f, sync_up_down = 2 | data_up_down = 1): one lane per frame, its array in LDS
  for (int fi = tid; fi < KT_BLOCK; fi += KT_WG)
    {
      const bool sync = fi < KT_SYNC;
      const int f = sync ? fi : fi - KT_SYNC;
      Stream st;
      st.seed (aes, (unsigned long long) f, sync ? 2 : 1);
      unsigned char *v = s_bands[tid];
      for (int i = 0; i < KT_NB; i++)
        v[i] = (unsigned char) (KT_MIN_BAND + i);
      // (the shuffle's steps 0 .. 59 settle the 60 places that are used; steps 60 .. 80 only move the rest among themselves: not drawn)
      for (int b = 0; b < 2 * KT_BPF / 2; b++)
        {
          unsigned long long d0, d1;
          st.block (aes, b, d0, d1);
          const int i0 = 2 * b, i1 = 2 * b + 1;
          {
            const int j = i0 + int (mod_small (d0, KT_NB - i0));
            const unsigned char x = v[i0]; v[i0] = v[j]; v[j] = x;
          }
          {
            const int j = i1 + int (mod_small (d1, KT_NB - i1));
            const unsigned char x = v[i1]; v[i1] = v[j]; v[j] = x;
          }
        }
      for (int i = 0; i < 60; i++)
        updown[size_t (fi) * 60 + i] = v[i];
    }
  __threadfence_block();
  __syncthreads();
  for (int i = tid; i < KT_MIX; i += KT_WG)               // (the band arrays are done with: the memory becomes the permutation)
    s_perm[i] = (unsigned short) i;
  /* Which swaps of the mix shuffle may go together is decided from the targets alone -- by all waves, before the sequential part:
   * a chunk of 64 steps is INDEPENDENT when every target lies beyond the chunk (or is the step's own place) and all targets differ
   * (the test is exact: 32 rotations of the targets through the wave compare every pair); the same for each quarter of 16 steps.
   * s_chunk[c]: bit 0 = the whole chunk is independent, bits 1 .. 4 = quarter q is NOT.  (In round 4 the one wave that applies the
   * swaps ran the tests itself, chunk by chunk: 0.59 ms of the kernel.) */
  constexpr int N_CHUNKS = (KT_MIX + 63) / 64;
  __shared__ __attribute__ ((aligned (4))) unsigned char s_chunk[(N_CHUNKS + 3) & ~3];
  {
    const int lane = tid & 63;
    for (int c = tid >> 6; c < N_CHUNKS; c += KT_WG / 64)
      {
        const int i_mine = c * 64 + lane, last = c * 64 + 63;
        const int j_mine = mix_t[min (i_mine, KT_MIX - 1)];
        const bool valid = i_mine < KT_MIX;
        // (a self swap keeps its place whatever the others do: it takes part with a target nobody else can have)
        const int j_cmp = j_mine == i_mine ? -1 - lane : j_mine;
        int others[32];
#pragma unroll
        for (int sft = 1; sft <= 32; sft++)                              // all rotations first, then the comparisons: no waits in between
          others[sft - 1] = __shfl (j_cmp, (lane + sft) & 63);
        int bad = (j_mine <= last) & (j_mine != i_mine);
#pragma unroll
        for (int sft = 0; sft < 32; sft++)
          bad |= others[sft] == j_cmp;
        const bool whole = !__any (bad != 0 && valid) && last < KT_MIX;
        int q_bad = (j_mine <= (i_mine | 15)) & (j_mine != i_mine);
#pragma unroll
        for (int sft = 1; sft <= 8; sft++)
          q_bad |= __shfl (j_cmp, (lane & ~15) | ((lane + sft) & 15)) == j_cmp;
        const unsigned long long bad_lanes = __ballot (q_bad != 0 || !valid);
        unsigned int flags = whole ? 1u : 0u;
#pragma unroll
        for (int q = 0; q < 4; q++)
          flags |= ((bad_lanes >> (16 * q)) & 0xffffull) ? 2u << q : 0u;
        if (lane == 0)
          s_chunk[c] = (unsigned char) flags;
      }
  }
  __syncthreads();

  // ---- the sequential swaps: three waves, one shuffle each.  The mix shuffle's targets lie in global memory: the wave fetches them a
  // few chunks ahead; an independent chunk is ONE LDS round trip, every lane its own swap; in a chunk that is not, an independent quarter
  // is one round trip of 16 lanes, and the others go four steps per round trip after a scalar test (the targets are taken out of the
  // registers with v_readlane -- a lane that loaded its own targets would wait a trip to memory per swap), else one by one.
  if (tid < 64)
    {
      static_assert (KT_MIX % 4 == 0, "the swaps are taken four at a time");
      // (the targets and flags of the NEXT FOUR chunks are on their way while four chunks are worked on: a chunk takes a few hundred
      // nanoseconds, its targets a microsecond to arrive)
      unsigned int t_ahead[4];
      unsigned int f_ahead = *reinterpret_cast<const unsigned int *> (s_chunk);
#pragma unroll
      for (int q = 0; q < 4; q++)
        t_ahead[q] = mix_t[min (q * 64 + tid, KT_MIX - 1)];
      for (int c0 = 0; c0 < N_CHUNKS; c0 += 4)
        {
          unsigned int t_now[4];
#pragma unroll
          for (int q = 0; q < 4; q++)
            t_now[q] = t_ahead[q];
          const unsigned int f_now = __builtin_amdgcn_readfirstlane (f_ahead);
#pragma unroll
          for (int q = 0; q < 4; q++)
            t_ahead[q] = mix_t[min ((c0 + 4 + q) * 64 + tid, KT_MIX - 1)];       // (clamped: no branch around a load in this loop)
          f_ahead = *reinterpret_cast<const unsigned int *> (s_chunk + min (c0 + 4, ((N_CHUNKS + 3) & ~3) - 4));
#pragma unroll
          for (int cq = 0; cq < 4; cq++)
            {
              const int c = c0 + cq;
              if (c >= N_CHUNKS)
                break;
              const unsigned int t = t_now[cq];
              const unsigned int flags = (f_now >> (8 * cq)) & 0xff;
              const int i_mine = c * 64 + tid;
              const int j_mine = int (t);
              if (flags & 1)
                {
                  const unsigned short x = s_perm[i_mine], y = s_perm[j_mine];
                  s_perm[i_mine] = y;
                  s_perm[j_mine] = x;
                  continue;
                }
              /* A chunk that is not independent as a whole: its four quarters of 16 steps one after the other, an independent quarter all
               * at once, else four steps per LDS round trip: steps i .. i + 3 with targets j0 .. j3 read { i + k, jk } and write the same
               * places; if no step reads what an earlier one of the four writes (jm != jk and jm != i + k for m < k; jk >= i + k > i + m
               * anyway), all eight reads can go out before the first write -- a scalar test, the targets are in scalar registers; a clash
               * takes the four steps one by one. */
#pragma unroll
              for (int q = 0; q < 4; q++)
                {
                  if (c * 64 + 16 * q >= KT_MIX)
                    continue;
                  if (!(flags & (2u << q)))
                    {
                      if ((tid >> 4) == q)
                        {
                          const unsigned short x = s_perm[i_mine], y = s_perm[j_mine];
                          s_perm[i_mine] = y;
                          s_perm[j_mine] = x;
                        }
                      continue;
                    }
#pragma unroll
                  for (int g4 = 0; g4 < 4; g4++)
                    {
                      const int g = 4 * q + g4;
                      const int i0 = c * 64 + 4 * g;
                      const int j0 = __builtin_amdgcn_readlane (int (t), 4 * g), j1 = __builtin_amdgcn_readlane (int (t), 4 * g + 1);
                      const int j2 = __builtin_amdgcn_readlane (int (t), 4 * g + 2), j3 = __builtin_amdgcn_readlane (int (t), 4 * g + 3);
                      if (i0 >= KT_MIX)
                        continue;
                      const bool clash = j0 == j1 || j0 == j2 || j0 == j3 || j1 == j2 || j1 == j3 || j2 == j3
                                      || j0 == i0 + 1 || j0 == i0 + 2 || j0 == i0 + 3 || j1 == i0 + 2 || j1 == i0 + 3 || j2 == i0 + 3;
                      if (tid == 0)
                        {
                          if (!clash)
                            {
                              const unsigned short a0 = s_perm[i0], a1 = s_perm[i0 + 1], a2 = s_perm[i0 + 2], a3 = s_perm[i0 + 3];
                              const unsigned short b0 = s_perm[j0], b1 = s_perm[j1], b2 = s_perm[j2], b3 = s_perm[j3];
                              s_perm[i0] = b0;     s_perm[j0] = a0;
                              s_perm[i0 + 1] = b1; s_perm[j1] = a1;
                              s_perm[i0 + 2] = b2; s_perm[j2] = a2;
                              s_perm[i0 + 3] = b3; s_perm[j3] = a3;
                            }
                          else
                            {
                              const int js[4] = { j0, j1, j2, j3 };
#pragma unroll
                              for (int k = 0; k < 4; k++)
                                {
                                  const unsigned short x = s_perm[i0 + k], y = s_perm[js[k]];
                                  s_perm[i0 + k] = y;
                                  s_perm[js[k]] = x;
                                }
                            }
                        }
                    }
                }
            }
        }
    }
  else if (tid == 64)
    apply_swaps (s_pos, s_pos_t, KT_BLOCK);
  else if (tid == 128)
    apply_swaps (s_order, s_order_t, KT_CODED);
  __syncthreads();

  if (GET)
    {
      clip_key_tables_tail (o, key, tid, s_perm, s_pos, s_order, updown);
      return;
    }
  // ---- the table: KEEP everywhere, then the bands of the sync frames and of the mix entries (wmcommon.cc build_frame_mod_table)
  signed char *table = a.tables + key * (long long) (2 * KT_BLOCK * KT_NB);
  {
    int *t4 = reinterpret_cast<int *> (table);
    for (int i = tid; i < 2 * KT_BLOCK * KT_NB / 4; i += KT_WG)
      t4[i] = 0;
    for (int i = (2 * KT_BLOCK * KT_NB / 4) * 4 + tid; i < 2 * KT_BLOCK * KT_NB; i += KT_WG)
      table[i] = 0;
  }
  __threadfence_block();
  __syncthreads();
  constexpr signed char UP = 1, DOWN = 2;
  // Both blocks (A, B) in one pass, eight entries per lane at a time with all their loads ahead of the stores: an entry is a chain of
  // dependent reads (permutation -> band bytes of its frame in global scratch, position, coded bit) in front of two byte stores, and one
  // workgroup per compute unit has nothing else to cover their latency with (one entry after the other: 0.69 of 2.1 ms per key).
  signed char *block_a = table, *block_b = table + size_t (KT_BLOCK) * KT_NB;
  const unsigned char *coded_a = a.coded, *coded_b = a.coded + KT_CODED;      // conv code of the payload, block type A / B
  constexpr int UB = 8;
  for (int e0 = tid; e0 < KT_SYNC * KT_BPF; e0 += KT_WG * UB)
    {
      int pos[UB], up[UB], down[UB], bit[UB];
#pragma unroll
      for (int k = 0; k < UB; k++)
        {
          const int e = min (e0 + k * KT_WG, KT_SYNC * KT_BPF - 1);
          const int f = e / KT_BPF, i = e % KT_BPF;
          bit[k] = (f / KT_SYNC_FPB) & 1;                                     // A carries 010101, B 101010
          pos[k] = s_pos[f];
          up[k] = updown[size_t (f) * 60 + i] - KT_MIN_BAND;
          down[k] = updown[size_t (f) * 60 + 30 + i] - KT_MIN_BAND;
        }
#pragma unroll
      for (int k = 0; k < UB; k++)
        if (e0 + k * KT_WG < KT_SYNC * KT_BPF)
          {
            block_a[pos[k] * KT_NB + up[k]] = bit[k] ? UP : DOWN;
            block_a[pos[k] * KT_NB + down[k]] = bit[k] ? DOWN : UP;
            block_b[pos[k] * KT_NB + up[k]] = bit[k] ? DOWN : UP;
            block_b[pos[k] * KT_NB + down[k]] = bit[k] ? UP : DOWN;
          }
    }
  for (int p0 = tid; p0 < KT_MIX; p0 += KT_WG * UB)
    {
      int pos[UB], up[UB], down[UB], bit_a[UB], bit_b[UB];
#pragma unroll
      for (int k = 0; k < UB; k++)
        {
          const int p = min (p0 + k * KT_WG, KT_MIX - 1);
          const int e = s_perm[p];                                            // entry (data frame f, i) that the shuffle put at position p
          const int f = e / KT_BPF, i = e % KT_BPF;
          const int o = s_order[p / (KT_BPF * KT_FPB)];                       // fec[p / 60], fec[k] = coded[order[k]]
          bit_a[k] = coded_a[o];
          bit_b[k] = coded_b[o];
          pos[k] = s_pos[KT_SYNC + f];
          up[k] = updown[size_t (KT_SYNC + f) * 60 + i] - KT_MIN_BAND;
          down[k] = updown[size_t (KT_SYNC + f) * 60 + 30 + i] - KT_MIN_BAND;
        }
#pragma unroll
      for (int k = 0; k < UB; k++)
        if (p0 + k * KT_WG < KT_MIX)
          {
            block_a[pos[k] * KT_NB + up[k]] = bit_a[k] ? UP : DOWN;
            block_a[pos[k] * KT_NB + down[k]] = bit_a[k] ? DOWN : UP;
            block_b[pos[k] * KT_NB + up[k]] = bit_b[k] ? UP : DOWN;
            block_b[pos[k] * KT_NB + down[k]] = bit_b[k] ? DOWN : UP;
          }
    }
}

__global__ void __launch_bounds__ (KT_WG)
frame_mod_table_kernel (KeyTableArgs a)
{
  key_tables_body<false> (a, ClipKeyTableOut {});
}

__global__ void __launch_bounds__ (KT_WG)
clip_key_table_kernel (KeyTableArgs a, ClipKeyTableOut o)
{
  key_tables_body<true> (a, o);
}

hipError_t
launch_frame_mod_tables (hipStream_t st, const KeyTableArgs& a)
{
  if (a.n_keys <= 0)
    return hipSuccess;
  if (!a.round_keys || !a.sbox || !a.coded || !a.scratch || !a.tables || a.scratch_slots <= 0 || a.n_keys > a.scratch_slots)
    return hipErrorInvalidValue;
  hipLaunchKernelGGL (frame_mod_table_kernel, dim3 ((unsigned) a.n_keys), dim3 (KT_WG), 0, st, a);
  return hipGetLastError();
}

hipError_t
launch_clip_key_tables (hipStream_t st, const KeyTableArgs& a, const ClipKeyTableOut& o)
{
  if (a.n_keys <= 0)
    return hipSuccess;
  if (!a.round_keys || !a.sbox || !a.scratch || a.scratch_slots <= 0 || a.n_keys > a.scratch_slots
      || !o.chains || !o.row_frames || !o.want || !o.perm || !o.pos || !o.mix_frame || !o.mix_up || !o.mix_down || !o.inv_order)
    return hipErrorInvalidValue;
  hipLaunchKernelGGL (clip_key_table_kernel, dim3 ((unsigned) a.n_keys), dim3 (KT_WG), 0, st, a, o);
  return hipGetLastError();
}

}  // namespace awmk
