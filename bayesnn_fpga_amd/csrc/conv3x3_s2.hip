// 3x3 / stride-2 / pad-1 convolution with the input patch resident in LDS as four PARITY PLANES (gfx950): the stride-2 convs of
// the path — BasicBlock conv1 of layers 2-4 and every exit-head conv, SA/models/resnet18/resnet18.py:280-299, :306-329 — on
// conv3x3_pw's 256 x 256 tile / 8 waves / ping-pong main loop, persistent like conv_igemm_wide_persist.
//
// Why.  conv_igemm_wide streams a [256 px][64 ch] activation tile per (tap, 64-channel chunk) from L2: every input line is
// requested 2.25 times per channel tile, and the re-reads miss the XCD's 4 MB L2 (32 workgroups x a 256-px tile's input, with
// the taps of one line 2-8 K-steps apart): rocprofv3 FETCH_SIZE 1.5x (64 -> 128 on 32x32 maps) to 4.2x (256 -> 512 + 512 on 8x8
// maps) the algorithmic input bytes, the kernel at 845 / 985 / 1070 TFLOP/s on the three shape classes against 1270-1420 for
// conv3x3_pw (profiles/experiments/r3_per_launch_before_s2.log).  Here a tile's input is DMA'd ONCE per 32-channel chunk and all
// nine taps are served from LDS.
//
// A stride-2 tap (ky, kx) reads input rows 2 oy + ky - 1 and columns 2 ox + kx - 1: rows of ONE parity per ky, columns of one
// parity per kx.  The patch is therefore kept as four planes — A (odd rows, odd columns; taps (0,0) (0,2) (2,0) (2,2)),
// B (odd, even; (0,1) (2,1)), C (even, odd; (1,0) (1,2)), D (even, even; (1,1)) — inside which a tap is a shift by 0 or 1 cell,
// exactly the stride-1 addressing of conv3x3_pw.  The taps run plane by plane (A A A A B B C C D per 32-channel chunk), so a
// plane's cells are dead long before they are needed again: ONE copy of the patch (72-88 KB for 32 channels) is enough, each
// plane is refilled with the next chunk's channels while the other planes compute (A during the C / D taps, B at the end of
// the period, C and D at the start of their own period), and nothing ever waits for a whole-patch refill.
//
//   tile       = IMGS whole output maps (256 pixels: 1 map of 16x16, 4 of 8x8, 16 of 4x4) x 256 output channels
//   waves      = 2 channel halves (g) x [2 (channels) x 2 (pixels)], wave tile 64 ch x 128 px = 4 x 8 tiles of v_mfma_f32_16x16x32;
//                the 16 pixels of an MFMA tile are a 4 x 4 block of output pixels (conv3x3_pw's tile-pixel order)
//   LDS        = [W0 | W1 (| W2) | patch pieces | (pad) | last stage] + BN table: three or four weight stages [256 ch][32 k] (64-byte rows), the patch as
//                NPT pieces of 128 cells x 64 B (piece = one DMA instruction per thread), planes back to back with plane A
//                padded to whole pieces.  132-152 KB: one 512-thread workgroup per CU.
//   64-B rows  = four rows share a 256-byte bank window.  ds_read_b128 is served in four lane groups {0-3, 12-15, 20-27},
//                {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS): a group reads 16-byte chunk kq of rows 0-3 and 12-15 of
//                a 16-row fragment and chunk kq ^ 1 of rows 4-11.  Chunk c of weight row r is stored at position
//                c ^ 2 ((r >> 2) & 1), chunk c of a patch cell in plane row y at c ^ 2 (y & 1): the 16 lanes of every group then
//                hit 16 distinct 16-byte slots, for every tap shift and every cell pitch (tight pitches: TW or TW + 1 cells).
//                LDS-DMA writes lane-linearly, so the permutation is applied to the per-lane SOURCE address.
//   main loop  = conv3x3_pw's: K-step = one tap x 32 channels, two phases (LOAD part / barrier / MFMA part / barrier), the two
//                wave groups one barrier apart, weights of the step AFTER NEXT in flight, counted vmcnt with compile-time
//                immediates (the 9 steps of a chunk are unrolled).  Per step at most two patch pieces ride along.
//   persistent = one workgroup per CU walks the tiles; the next tile's weight stages 0 / 1 and its plane A / B pieces are
//                issued when the main loop ends and land while the epilogue runs (BN + ReLU on the accumulators, fp16 straight from the
//                registers: S2_DIRECT; formerly two rounds of 128 pixels per channel half through the 64 KB the C / D pieces and W2 occupy).
//   pair mode  = as conv_igemm_wide: channel tiles >= split use the second conv's weights / BN / output tensor.
//   epilogue   = plain (BN + ReLU) only: every stride-2 conv of the path.  Anything else stays with conv_igemm_wide.
#include <cstdlib>

#include "conv_epilogue.h"
#include "kernels.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#ifndef S2_DIRECT
#define S2_DIRECT 1          // conv3x3_pwp's PWP_DIRECT: the MFMA rows of a wave's four channel tiles are a permutation of its 64 channels (row r of tile i =
                             // channel 32 (i >> 1) + 8 (r >> 2) + 4 (i & 1) + (r & 3)), so a lane holds two runs of 8 consecutive channels per pixel and the
                             // epilogue stores them straight from the registers — no trip through LDS, no barrier.  0: the two rounds through LDS
#endif
#ifndef S2_RELAX0
#define S2_RELAX0 0          // 1: the previous tile's 16 output stores may stay in flight through step 0 of the next tile (see END_OF_STEP_WAIT).
                             // Built and measured in round 4 (same-box A/B on the five stride-2 classes, bit-identical): neutral (+-0.5 %): off
#endif
#ifndef S2_DRAIN_STORES
#define S2_DRAIN_STORES 0    // 1 (diagnostic): every tile starts with vmcnt(0)
#endif
#ifndef S2_WSPLIT
#define S2_WSPLIT 0          // 1: a wave issues its two weight DMA instructions of a K-step in the two phases (one each) instead of both in phase 0
#endif
#ifndef S2_STORE_SC1
#define S2_STORE_SC1 0       // 1: the output tile leaves with write-through stores that do not stay in the XCD's L2 (sc1, inline asm).  A 32-channel
                             // chunk is half (a quarter) of an input pixel's 128-byte lines, whose other half is asked for 9 K-steps later, and
                             // the 128 KB of output per tile push those lines out of the 4 MB L2 in between: with sc1 stores rocprofv3 FETCH_SIZE
                             // over the six stride-2 launches of the headline step went 16.3 -> 14.3 GB (conv_igemm_wide: 17.9) at UNCHANGED
                             // wall time (the re-fetches are Infinity-Cache hits; profiles/experiments/r3_s2_sc1_stores.log).  OFF: hipcc does
                             // not model an asm store (cdna_hip_programming.md 5.7 item 1); the first build without the trailing s_nop failed
                             // every model parity test, and with it one build still gave ONE image of one dynamic-exit test a 1e-6 deviation
                             // that a recompile with an unrelated edit removed — a data-register hazard that depends on instruction
                             // placement is not worth a counter that does not show up in the time.
#endif
#ifndef S2_PIECE_PHASE
#define S2_PIECE_PHASE 1     // phase of a K-step whose LOAD part issues the step's patch pieces (phase 0 carries 8 fragment reads and
                             // the 2 weight DMAs, phase 1 only 4 reads)
#endif
#ifndef S2_NST_BIG
#define S2_NST_BIG 3         // weight stages on the 32x32 / 16x16 input maps (3 | 4: four measured 1-2 % slower, below; the 8x8 maps have LDS for 3)
#endif

// Timing probes (tools/ab_build.py name:-DS2_ABL_...=1; wrong results by construction, never in the product build): which part of a
// tile's life the time goes to.
#ifndef S2_ABL_NOPATCH
#define S2_ABL_NOPATCH 0     // no patch DMA (the main loop reads whatever is in LDS)
#endif
#ifndef S2_ABL_NOSTORE
#define S2_ABL_NOSTORE 0     // the epilogue runs but stores nothing
#endif
#ifndef S2_ABL_HALFREADS
#define S2_ABL_HALFREADS 0   // every second pixel-fragment read is skipped (8 instead of 12 reads per 32 MFMAs)
#endif
#ifndef S2_ABL_NOBITS
#define S2_ABL_NOBITS 0      // masked-input form: no keep-bit DMA (the masking reads whatever the slots hold)
#endif
#ifndef S2_ABL_NORMW
#define S2_ABL_NORMW 0       // masked-input form: the pieces are not masked in LDS
#endif
#ifndef S2_MASK_PHASE
#define S2_MASK_PHASE 0      // masked-input form: phase of a K-step whose LOAD part masks the pieces that are due
#endif
#ifndef S2_BITS_GROUP
#define S2_BITS_GROUP 0      // 1: ONE keep-bit DMA serves four pieces (the four lanes of a cell fetch the dwords of four different pieces): three
                             // bit DMAs per chunk instead of ten.  Measured SLOWER (2.21 -> 2.37 ms on the headline's pair launch): the per-lane
                             // choice among four piece offsets costs 7 VGPRs the kernel does not have, and two of the spilled values are
                             // reloaded inside the main loop (a scratch load in front of a DMA is a vmcnt wait for everything in flight)
#endif
#ifndef S2_MASK_LUT
#define S2_MASK_LUT 1        // 1: the four dword masks of a keep byte come from a 256-entry table in LDS (one ds_read_b128) instead of 16 vector
                             // instructions; with S2_MASK_ATOMIC the pair launch of the headline 2.245 -> 2.18 ms (r3_lazy_site.log)
#endif
#ifndef S2_MASK_ATOMIC
#define S2_MASK_ATOMIC 1     // 1: the dropped elements are cleared by two ds_and_b64 (the LDS unit reads, ANDs and writes) instead of
                             // ds_read_b128 / v_and / ds_write_b128 through the registers
#endif
#ifndef S2_ABL_NOEPI
#define S2_ABL_NOEPI 0       // no epilogue at all (one accumulator element per lane is stored so the MFMAs stay live)
#endif

// TW = OUTPUT map size (TW x TW; the input map is 2 TW x 2 TW): 16, 8 or 4.
// NST = weight stages: the weights of K-step T + NST - 1 are issued in step T.  vmcnt retires in order, so waiting for the NEXT
// step's weights also waits for every patch piece issued before them: a piece issued in step s must have landed by the end of
// step s + NST - 1 whether its plane is needed by then or not.  With 3 stages that is 2 K-steps (~1.4 us) for an HBM round
// trip.  Timing probes (profiles/experiments/r3_s2_ablation.log) showed the 32x32 / 16x16 classes 15-20 % faster WITHOUT the
// patch DMA, which suggested that window; a fourth stage (one more K-step, built and kept behind -DS2_NST_BIG=4) measured
// 1-2 % SLOWER on all three classes (same-box A/B, r3_s2_stages_phase.log): the window is not what they wait for.
// MSK = masked input (ConvArgs::in_bits): the patch is DMA'd from the deterministic (pre-scaled) tensor, a piece's keep bits ride
// along as one more DMA (a dword per lane into a 2 KB slot), and the thread that issued a piece clears the dropped elements of
// ITS 16 bytes in LDS one K-step after the piece has landed — a piece is readable one step later than without the mask.
// CT_ = channels per tile: 256, or 128 (round 4, 16x16 output maps only: ResNet-50's Cout = 128 stride-2 convs).  With 128 the two wave groups
// split the tile's PIXELS (output rows 0-7 / 8-15) instead of its channels: a wave tile is 64 ch x 64 px (4 x 4 MFMA tiles), both groups read the
// same 8 KB weight stage, a K-step's two phases take two pixel blocks each (8 MFMAs per phase instead of 16; the barrier schedule, the patch and
// its refill windows are the same), and a group's 128 ch x 128 px are ONE round of the epilogue.
template <int TW, int NST_ = (TW == 4 ? 3 : S2_NST_BIG), bool MSK_ = false, int CT_ = 256>
struct S2Geom {
    static_assert(CT_ == 256 || (CT_ == 128 && TW == 16), "128-channel tiles: the 16x16 output maps");
    static constexpr int NST = NST_;
    static constexpr bool MSK = MSK_;
    static constexpr int LAND = NST + (MSK ? 1 : 0);              // a piece issued in step s is readable from step s + LAND on
    static_assert(NST == 3 || NST == 4, "weight stages");
    static constexpr int BN_MAX = TW == 4 ? 1024 : 512;           // channels of the launch (both convs of a pair): the BN table
    static constexpr int CT = CT_, PX = 256, IMGS = PX / (TW * TW);
    static constexpr int WPI = CT / 128;                          // weight DMA instructions per thread and K-step
    static_assert(IMGS >= 1 && IMGS * TW * TW == PX, "whole output maps per tile");
    // planes in tap-use order: 0 = A (odd input rows, odd columns), 1 = B (odd, even), 2 = C (even, odd), 3 = D (even, even).
    // Plane row y of an odd-row plane is input row 2 y - 1 (y = 0: the padding row), of an even-row plane input row 2 y.
    __host__ __device__ static constexpr int py(int pl) { return pl < 2 ? 1 : 0; }
    __host__ __device__ static constexpr int px(int pl) { return (pl == 0 || pl == 2) ? 1 : 0; }
    __host__ __device__ static constexpr int rows(int pl) { return TW + py(pl); }
    __host__ __device__ static constexpr int cols(int pl) { return TW + px(pl); }                 // = cell pitch
    __host__ __device__ static constexpr int cells(int pl) { return IMGS * rows(pl) * cols(pl); }
    static constexpr int A_PAD = (cells(0) + 127) / 128 * 128;                                    // plane A ends on a piece boundary
    __host__ __device__ static constexpr int cell0(int pl) {
        return pl == 0 ? 0 : A_PAD + (pl >= 2 ? cells(1) : 0) + (pl >= 3 ? cells(2) : 0);
    }
    static constexpr int TOTAL = cell0(3) + cells(3);
    static constexpr int NPT = (TOTAL + 127) / 128;               // pieces (= DMA instructions per thread and chunk): 11 | 10 | 10
    __host__ __device__ static constexpr int plane_of(int cell) { return cell < A_PAD ? 0 : (cell < cell0(2) ? 1 : (cell < cell0(3) ? 2 : 3)); }
    __host__ __device__ static constexpr int lo(int k) { return plane_of(128 * k); }
    __host__ __device__ static constexpr int hi(int k) { return plane_of(128 * k + 127 < TOTAL ? 128 * k + 127 : TOTAL - 1); }
    __host__ __device__ static constexpr int count_hi_le(int pl) { int n = 0; for (int k = 0; k < NPT; ++k) n += hi(k) <= pl ? 1 : 0; return n; }
    static constexpr int NA = count_hi_le(0);                     // pieces that hold plane A only
    static constexpr int PRO = count_hi_le(1);                    // ... planes A / B only: loaded ahead (prologue, next chunk)
    static constexpr int NB = PRO - NA;
    static constexpr int NOWN = NPT - PRO;                        // pieces with C / D cells: loaded in their own chunk's period
    // K-step (0..8 of a chunk's period) in which piece k is issued.  A pieces (next chunk): steps 5, 6 (step 5 with four
    // stages: they must have landed NST - 1 steps later, before step 0); B pieces (next chunk): steps 7, 8; own-period pieces:
    // one per step from step 0.
    __host__ __device__ static constexpr int pstep(int k) {
        return k < NA ? (NST == 3 && !MSK ? 5 + (2 * k) / NA : 5) : (k < PRO ? 7 + (2 * (k - NA)) / NB : k - PRO);
    }
    __host__ __device__ static constexpr int pieces_at(int s) { int n = 0; for (int k = 0; k < NPT; ++k) n += pstep(k) == s ? 1 : 0; return n; }
    // Validity of that schedule (see the hazard notes in the kernel): a plane last read in step L may be overwritten from
    // step L + 2 on; a piece issued in step s has landed for every wave at the end of step s + NST - 1.
    __host__ __device__ static constexpr bool schedule_ok() {
        constexpr int first[4] = {0, 4, 6, 8}, last[4] = {3, 5, 7, 8};
        for (int k = 0; k < NPT; ++k) {
            const int s = pstep(k);
            if (k < PRO) {            // carries the NEXT chunk: period steps 9 + first[lo] is the deadline
                if (s < last[hi(k)] + 2 || s > 8 || s + LAND > 9 + first[lo(k)]) return false;
            } else {                  // carries its own chunk; the plane was last read in the previous period
                if (s + 9 < last[hi(k)] + 2 || s + LAND > first[lo(k)] || s > 4) return false;
            }
        }
        for (int s = 0; s < 9; ++s)
            if (pieces_at(s) > 3) return false;
        return NOWN <= 5 && NA >= 1 && NB >= 1;
    }
    static_assert(schedule_ok(), "patch refill schedule violates a WAR / RAW window");
    // DMA instructions of step s of a chunk (`last`: the tile's last chunk — no next-chunk pieces, no weights beyond the tile)
    __host__ __device__ static constexpr bool w_issued(int s, bool last) { return s + NST - 1 <= 8 || !last; }
    // keep-bit DMAs (MSK).  Grouped: pieces 0-3 (A, A, A, B: next chunk) at step 5, pieces 4 (B, next chunk) + 9 (own) at step 4, pieces 5-8
    // (own) at step 0 — each with its group's earliest piece, so a piece's bits land no later than the piece.
    __host__ __device__ static constexpr int grp(int k) { return k < 4 ? 0 : (k == 4 || k == 9 ? 1 : 2); }
    __host__ __device__ static constexpr int gidx(int k) { return k < 4 ? k : (k == 4 ? 0 : (k == 9 ? 1 : k - 5)); }
    __host__ __device__ static constexpr int bits_at(int s, bool last) {
        if (!MSK || S2_ABL_NOBITS) return 0;
        if (!S2_BITS_GROUP) return (s <= 4 || !last) ? pieces_at(s) : 0;
        return (s == 0 || s == 4) ? 1 : ((s == 5 && !last) ? 1 : 0);
    }
    __host__ __device__ static constexpr int p_issued(int s, bool last) { return ((s <= 4 || !last) ? pieces_at(s) : 0) + bits_at(s, last); }
    // What may still be in flight when step S ends: everything issued BEHIND the weights of step S + 1 (which were issued first
    // thing in step S - (NST - 2)): that step's pieces, then weights + pieces of the steps up to S.  (Steps before 0 are the
    // previous chunk's, never a last one; in a tile's first chunk they do not exist and the count is merely generous: the
    // prologue has waited for everything the first NST - 1 steps read.)
    __host__ __device__ static constexpr int wait_n(int S, bool last) {
        int n = 0;
        for (int j = S - (NST - 2); j <= S; ++j) {
            const bool prev = j < 0;
            const int sj = prev ? j + 9 : j;
            const bool lastj = last && !prev;
            if (j > S - (NST - 2)) n += w_issued(sj, lastj) ? WPI : 0;
            n += p_issued(sj, lastj);
        }
        return n;
    }
    static constexpr int WST = CT * 64;                           // one weight stage (32-deep K-step)
    static constexpr int PIECE = 512 * 16;
    static constexpr int P_OFF = (NST - 1) * WST;                 // [W0 | W1 (| W2) | pieces ... | (pad) | last stage | BN]
    static constexpr int E_OFF = P_OFF + PRO * PIECE;             // epilogue staging: the own-period pieces, padding, the last stage
    static constexpr int E_BYTES = 65536;
    static constexpr int END_PIECES = P_OFF + NPT * PIECE;
    static constexpr int WL_OFF = (END_PIECES + WST > E_OFF + E_BYTES ? END_PIECES : E_OFF + E_BYTES - WST);
    static constexpr int BN_OFF = WL_OFF + WST;
    static constexpr int BITS_OFF = BN_OFF + 2 * BN_MAX * 4 + 128;      // (behind the two image-row tables of the dynamic-exit instantiation)
    // keep-bit slots (2 KB: a dword per lane) of the pieces in flight.  A slot is live from the piece's issue to its masking three
    // steps later: own-period pieces (steps 0-4 -> 3-7) take slots 0-4, the A pieces (step 5 -> 8) slots 0, 1, 5, the B pieces
    // (steps 7, 8 -> 1, 2 of the next period) slots 2, 3.
    __host__ __device__ static constexpr int first_due(int S) {       // first piece whose masking is due in step S (issued in step S - 3 or S + 6), or -1
        for (int k = 0; k < NPT; ++k)
            if (pstep(k) + 3 == S || pstep(k) + 3 == S + 9) return k;
        return -1;
    }
    static constexpr int NSLOT = MSK ? (S2_BITS_GROUP ? 3 : 6) : 0;
    __host__ __device__ static constexpr int slot(int k) { return S2_BITS_GROUP ? grp(k) : (k >= PRO ? k - PRO : (k < NA ? (k < 2 ? k : 5) : 2 + (k - NA))); }
    static_assert(!MSK || (NA == 3 && NB == 2 && NOWN == 5 && NPT == 10), "keep-bit slot plan");
    static constexpr int LUT_OFF = BITS_OFF + NSLOT * 2048;
    static constexpr int LDS_BYTES = LUT_OFF + (MSK && S2_MASK_LUT ? 4096 : 0);
    static_assert(BN_OFF - E_OFF >= E_BYTES, "epilogue staging area");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    __host__ __device__ static constexpr int wstage_off(int st) { return st == NST - 1 ? WL_OFF : st * WST; }
    // tile pixel p -> image of the tile and output coordinates: the 16 pixels of one MFMA tile are a 4 x 4 block
    __host__ __device__ static constexpr int p_img(int p) { return p / (TW * TW); }
    __host__ __device__ static constexpr int p_ox(int p) { return 4 * ((p >> 4) % (TW / 4)) + (p & 3); }
    __host__ __device__ static constexpr int p_oy(int p) { return 4 * (((p >> 4) / (TW / 4)) % (TW / 4)) + ((p >> 2) & 3); }
    static constexpr int BR = TW / 4, BI = BR * BR;
    // cell of the j-th 4 x 4 block of a wave (its 8 blocks start at a multiple of 8) relative to the wave's first one, plane pl
    __host__ __device__ static constexpr int cell_delta(int pl, int j) {
        return (j / BI) * rows(pl) * cols(pl) + 4 * ((j % BI) / BR) * cols(pl) + 4 * ((j % BI) % BR);
    }
};

// step s of a chunk's period -> plane, cell shift (dy, dx) inside the plane, tap index ky * 3 + kx (weight offset)
__host__ __device__ constexpr int s2_plane(int s) { return s < 4 ? 0 : (s < 6 ? 1 : (s < 8 ? 2 : 3)); }
__host__ __device__ constexpr int s2_dy(int s) { return (s == 2 || s == 3 || s == 5) ? 1 : 0; }
__host__ __device__ constexpr int s2_dx(int s) { return (s == 1 || s == 3 || s == 7) ? 1 : 0; }
__host__ __device__ constexpr int s2_tap(int s) {
    constexpr int t[9] = {0, 2, 6, 8, 1, 7, 3, 5, 4};
    return t[s];
}

template <int TW, bool BF, bool IMAP, bool MSK = false, int CT_ = 256>
__global__ __launch_bounds__(512, 1) void conv3x3_s2_kernel(ConvArgs a, int n_tiles) {
    using G = S2Geom<TW, (TW == 4 ? 3 : S2_NST_BIG), MSK, CT_>;
    static_assert(!MSK || (TW == 16 && !IMAP), "masked input: one image per tile, the ordinary form");
    constexpr int CT = G::CT, IMGS = G::IMGS, NPT = G::NPT, PRO = G::PRO, NST = G::NST;
    constexpr bool HALF = CT == 128;
    constexpr int TI = 4, TP = HALF ? 4 : 8, JB = TP / 2;        // JB pixel blocks per phase of a K-step
    typedef float accv __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) char smem[G::LDS_BYTES];
    char* const pbuf = smem + G::P_OFF;
    float* const bn_scale = (float*)(smem + G::BN_OFF);
    float* const bn_bias = bn_scale + G::BN_MAX;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, kq = lane >> 4;
    const int g = wave >> 2, wc = (wave >> 1) & 1, wp = wave & 1;

    const int n_ctiles = a.Cout / CT;
    const int n_ptiles = (a.N + IMGS - 1) / IMGS;
    const int Ktot = 9 * a.Cin;
    const int nC = a.Cin / 32;
    const int split = a.wgt_b ? a.split : a.Cout;

    // folded-BN table of the whole launch (an ordinary global load inside the epilogue would make hipcc drain the LDS-DMA)
    for (int c = tid; c < a.Cout; c += 512) {
        const bool second = c >= split;
        const float* sp = second ? a.scale_b : a.scale;
        const float* bp = second ? a.bias_b : a.bias;
        const int cc = second ? c - split : c;
        bn_scale[c] = (sp ? sp[cc] : 1.f) * a.out_mul;
        bn_bias[c] = bp ? bp[cc] : 0.f;
    }
    if constexpr (MSK && S2_MASK_LUT) {
        if (tid < 256) {
            typedef unsigned int u32x4_l __attribute__((ext_vector_type(4)));
            u32x4_l m;
#pragma unroll
            for (int i = 0; i < 4; ++i) m[i] = (((tid >> (2 * i)) & 1) ? 0xffffu : 0u) | (((tid >> (2 * i + 1)) & 1) ? 0xffff0000u : 0u);
            *(u32x4_l*)(smem + G::LUT_OFF + tid * 16) = m;
        }
    }
    __syncthreads();   // (no LDS-DMA in flight yet: the plain barrier and its waits are fine here)

    // ---- per-lane fragment geometry (tile-independent) ----
    // weights: row (g*128 + wc*64 + l16 + 16 i), chunk kq at position kq ^ 2 ((row >> 2) & 1) = kq ^ 2 ((l16 >> 2) & 1)
    const int a_off = ((HALF ? 0 : g * 128) + wc * 64 + (S2_DIRECT ? 8 * (l16 >> 2) + (l16 & 3) : l16)) * 64 + ((kq ^ (((l16 >> 2) & 1) << 1)) << 4);
    // byte offset of channel tile i's row of this lane relative to a_off
#define A_TILE(I) (S2_DIRECT ? (32 * ((I) >> 1) + 4 * ((I) & 1)) * 64 : (I) * 16 * 64)
    // pixels: lane (by, bx) = (l16 >> 2, l16 & 3) of each 4 x 4 block; plane row parity of the cell = (by + dy) & 1
    const int by = l16 >> 2, bx = l16 & 3;
    const int pbase = HALF ? (g * 2 + wp) * 64 : wp * 128;        // first tile pixel of this wave
    // (a tap with dy = 1 reads the cell one plane row further down: + cols * 64 as an immediate, and the chunk sits at the
    // other position of its pair: the same per-lane address with bit 5 flipped)
    int boff[4];
#pragma unroll
    for (int pl = 0; pl < 4; ++pl) {
        const int wave_cell = G::cell0(pl) + (G::p_img(pbase) * G::rows(pl) + G::p_oy(pbase)) * G::cols(pl) + G::p_ox(pbase);
        boff[pl] = G::P_OFF + (wave_cell + by * G::cols(pl) + bx) * 64 + ((kq ^ ((by & 1) << 1)) << 4);
    }

    // ---- per-thread DMA sources ----
    // Every DMA is a buffer_load ... lds through a per-tile buffer descriptor (wave-uniform, SGPRs) with a 32-bit per-lane byte
    // offset: no 64-bit address arithmetic or pointer selects in the loop, and a lane whose offset lies beyond the descriptor's
    // size loads ZEROS — the padding ring of the patch, the cells behind a plane, the images of a last tile beyond N.
    // patch: piece q = tid + 512 k -> cell q >> 2 of the piece array, position q & 3 holds chunk (q & 3) ^ 2 (y & 1).  What a
    // piece reads is the same for every tile up to the tile's first image: pre[k] = byte offset of the piece's source
    // relative to image n0 of the input (2 (img * H*W*Cin + (iy * W + ix) * Cin + chunk * 8)), computed ONCE per kernel; the
    // descriptor of a tile starts at its image n0 and ends behind its last image.  (IMAP: the images of a tile are not
    // consecutive rows — pre[k] = img << 24 | in-image ELEMENT offset, the rows come from a per-tile LDS table and the
    // descriptor covers the whole tensor.)
    const unsigned HWC = (unsigned)a.H * a.W * a.Cin;             // (launcher: N <= in_mod, in_mod * HWC < 2^31, HWC < 2^24)
    constexpr unsigned OOB = 0xfffffff0u;                         // beyond every descriptor (a tile's <= 16 images; IMAP: the tensor, < 2^32 - 16 bytes)
    unsigned pre[NPT];
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int q = tid + 512 * k;
        const int cell = q >> 2, pos = q & 3;
        const int pl = G::lo(k) == G::hi(k) ? G::lo(k) : (cell < G::cell0(G::hi(k)) ? G::lo(k) : G::hi(k));
        const int pyl = pl < 2 ? 1 : 0, pxl = (pl == 0 || pl == 2) ? 1 : 0;
        const int RQ = (TW + pyl) * (TW + pxl), Q = TW + pxl;
        const int lc = cell - (pl == 0 ? 0 : (pl == 1 ? G::cell0(1) : (pl == 2 ? G::cell0(2) : G::cell0(3))));
        const int img = lc / RQ, rm = lc - img * RQ;
        const int y = rm / Q, x = rm - y * Q;
        const int iy = 2 * y - pyl, ix = 2 * x - pxl;
        const bool ok = lc < IMGS * RQ && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        // (masked input in the lazy site's planar layout, kernels.h lazy_planar_off: a plane row's cells are contiguous 64-byte runs)
        const unsigned inoff = (MSK && a.lazy_planar) ? (unsigned)((iy * a.W + (ix & 1) * (a.W >> 1) + (ix >> 1)) * 32 + ((pos ^ ((y & 1) << 1)) << 3))
                                                      : (unsigned)((iy * a.W + ix) * a.Cin + ((pos ^ ((y & 1) << 1)) << 3));
        pre[k] = !ok ? OOB : (IMAP ? ((unsigned)img << 24) | inoff : 2u * ((unsigned)img * HWC + inoff));
    }
    // IMAP: tensor row of each image of a tile (-1 beyond N), two tables: the epilogue of tile i reads table i & 1 while the
    // prologue of tile i + 1 is issued from table (i + 1) & 1
    int* const row_tabs = (int*)(smem + G::BN_OFF + 2 * G::BN_MAX * 4);
    int tsel = 0;
    // weights: piece q = tid + 512 i -> row (tid >> 2) + 128 i, position tid & 3 holds chunk (tid & 3) ^ 2 ((row >> 2) & 1);
    // one descriptor per 128-row half of the channel tile (a pair's second conv has its own weight tensor)
    // (S2_DIRECT: a fragment's 16 lanes read rows 8 a + b + const: the chunk position alternates with row >> 3 instead of row >> 2)
    const unsigned woff = 2u * ((unsigned)(tid >> 2) * Ktot + (((tid & 3) ^ (((tid >> (S2_DIRECT ? 5 : 4)) & 1) << 1)) << 3));
    const unsigned wbytes = 2u * 128u * Ktot;
    // element offset of channel chunk C0 (a multiple of 32) = C0 * cmul: 1 in NHWC, H * W in the planar layout (one 32-channel plane per chunk)
    const unsigned cmul = (MSK && a.lazy_planar) ? (unsigned)(a.H * a.W) : 1u;
    __amdgpu_buffer_rsrc_t rs_w0, rs_w1, rs_in, rs_bits;

    int ch0 = 0, n0 = 0;

    // (the K offset of a step — tap and channel chunk, wave-uniform — rides in the instruction's SGPR offset: the per-lane
    // offsets stay loop-invariant, one VGPR each, and nothing is added per DMA)
#define BLDS16(RSRC, VOFF, SOFF, LDSPTR) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds((RSRC), (__attribute__((address_space(3))) void*)(LDSPTR), 16, (VOFF), (SOFF), 0, 0)
#define ISSUE_W_HALF(KOFF, ST, I)                                                                            \
    {                                                                                                        \
        const unsigned so_ = __builtin_amdgcn_readfirstlane(2u * (unsigned)(KOFF));                          \
        if ((I) == 0) BLDS16(rs_w0, woff, so_, smem + G::wstage_off(ST) + (0 * 512 + wave * 64) * 16);       \
        else BLDS16(rs_w1, woff, so_, smem + G::wstage_off(ST) + (1 * 512 + wave * 64) * 16);                \
    }
#define ISSUE_W(KOFF, ST)                                                                                    \
    {                                                                                                        \
        ISSUE_W_HALF(KOFF, ST, 0);                                                                           \
        if (!HALF) ISSUE_W_HALF(KOFF, ST, 1);                                                                \
    }
#define ISSUE_P(K, C0)                                                                                       \
    {                                                                                                        \
        unsigned o_ = pre[K];                                                                                \
        if constexpr (IMAP) {                                                                                \
            const int row_ = row_tabs[tsel * 16 + ((pre[K] >> 24) & 15)];                                    \
            o_ = (pre[K] == OOB || row_ < 0) ? OOB : 2u * ((unsigned)row_ * HWC + (pre[K] & 0xffffffu));     \
        }                                                                                                    \
        if (!S2_ABL_NOPATCH) BLDS16(rs_in, o_, __builtin_amdgcn_readfirstlane(2u * (unsigned)(C0) * cmul), pbuf + ((K) * 512 + wave * 64) * 16); \
        if constexpr (MSK && !S2_ABL_NOBITS && !S2_BITS_GROUP) {                                             \
            /* the cell's 32 keep bits of this chunk (the dword that holds this piece's byte); beyond the descriptor: zeros */ \
            /* (derived from pre[K] at every issue: as loop invariants the ten offsets are spilled, and a scratch reload in */ \
            /*  front of a DMA is a vmcnt wait for everything in flight)                                                    */ \
            unsigned bo_ = pre[K];                                                                           \
            asm volatile("" : "+v"(bo_));                                                                    \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_bits, (__attribute__((address_space(3))) void*)(smem + G::BITS_OFF + G::slot(K) * 2048 + wave * 256), \
                                                     4, (bo_ >> 4) & ~3u, __builtin_amdgcn_readfirstlane(((unsigned)(C0) * cmul) >> 3), 0, 0); \
        }                                                                                                    \
    }
    // Grouped keep-bit DMA: lane (cell, pos) fetches the dword of piece PA / PB / PC / PD (pos 0..3; -1: none) of ITS cell row; CB = channel
    // offset of the group's chunk (bytes of bits: / 8), ADD0 = extra bytes for pos 0 (the one piece of the mixed group that carries the
    // next chunk).  The four lanes of a cell are lanes of one wave: the wave's vmcnt wait covers what its neighbours fetched.
#define ISSUE_BITS(GRP, PA, PB, PC, PD, CB, ADD0)                                                            \
    if constexpr (MSK && !S2_ABL_NOBITS && S2_BITS_GROUP) {                                                  \
        int p4_ = tid & 3;                                                                                   \
        asm volatile("" : "+v"(p4_));                                                                        \
        unsigned bo_ = p4_ == 0 ? pre[PA] : (p4_ == 1 ? pre[PB] : ((PC) >= 0 ? (p4_ == 2 ? pre[(PC) < 0 ? 0 : (PC)] : pre[(PD) < 0 ? 0 : (PD)]) : OOB)); \
        bo_ = ((bo_ >> 4) & ~3u) + (p4_ == 0 ? (unsigned)(ADD0) : 0u);                                       \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_bits, (__attribute__((address_space(3))) void*)(smem + G::BITS_OFF + (GRP) * 2048 + wave * 256), \
                                                 4, bo_, __builtin_amdgcn_readfirstlane(((unsigned)(CB) * cmul) >> 3), 0, 0); \
    }
    // The thread that issued piece K clears the dropped elements of its 16 bytes: byte (logical chunk) of the slot's dword -> four
    // dword masks.  Own DMA only: the thread's counted vmcnt wait is all it needs; the barriers of the step publish the result.
    typedef unsigned int u32x4_m __attribute__((ext_vector_type(4)));
#define MASK_LOAD(K, W, V)                                                                                   \
    if (!S2_ABL_NORMW) {                                                                                     \
        int t4_ = S2_BITS_GROUP ? ((tid & ~3) + G::gidx(K)) * 4 : tid * 4;                                   \
        asm volatile("" : "+v"(t4_));                                                                        \
        W = *(const unsigned*)(smem + G::BITS_OFF + G::slot(K) * 2048 + t4_);                                /* the dword of this piece's cell */ \
        if (!S2_MASK_ATOMIC) V = *(const u32x4_m*)(pbuf + ((K) * 512 + tid) * 16);                           \
    }
#define MASK_STORE(K, W, V)                                                                                  \
    if (!S2_ABL_NORMW) {                                                                                     \
        unsigned sh_ = pre[K];                                                                               \
        asm volatile("" : "+v"(sh_));                                                                        \
        const int b_ = (int)((W) >> ((sh_ >> 1) & 24u));                                                     \
        u32x4_m v_ = V, m_;                                                                                  \
        if (S2_MASK_LUT) {                                                                                   \
            m_ = *(const u32x4_m*)(smem + G::LUT_OFF + ((b_ & 0xff) << 4));                                  \
            v_ &= m_;                                                                                        \
        } else {                                                                                             \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                               \
                const unsigned lo_ = (unsigned)__builtin_amdgcn_sbfe(b_, 2 * i_, 1), hi_ = (unsigned)__builtin_amdgcn_sbfe(b_, 2 * i_ + 1, 1); \
                m_[i_] = (lo_ & 0xffffu) | (hi_ & 0xffff0000u);                                              \
                v_[i_] &= m_[i_];                                                                            \
            }                                                                                                \
        }                                                                                                    \
        if (S2_MASK_ATOMIC) {                                                                                \
            typedef __attribute__((address_space(3))) unsigned long long lds_u64;                            \
            lds_u64* const q_ = (lds_u64*)(pbuf + ((K) * 512 + tid) * 16);                                   \
            __atomic_fetch_and(q_, ((unsigned long long)m_[1] << 32) | m_[0], __ATOMIC_RELAXED);             \
            __atomic_fetch_and(q_ + 1, ((unsigned long long)m_[3] << 32) | m_[2], __ATOMIC_RELAXED);         \
        } else {                                                                                             \
            *(u32x4_m*)(pbuf + ((K) * 512 + tid) * 16) = v_;                                                 \
        }                                                                                                    \
    }
#define MASK_P(K)                                                                                            \
    {                                                                                                        \
        unsigned mw_;                                                                                        \
        u32x4_m mv_;                                                                                         \
        MASK_LOAD(K, mw_, mv_);                                                                              \
        MASK_STORE(K, mw_, mv_);                                                                             \
    }
    // piece whose masking is due in step S of a period (issued three steps earlier), -1: none.  (Step 8: the A pieces, all three
    // issued in step 5: the first one takes the early-load slot, the others follow serially.)
#define MASK_DUE(S, K) (G::pstep(K) + 3 == (S) || G::pstep(K) + 3 == (S) + 9)
    // Tile VB: its buffer descriptors; its first two weight stages and the pieces that hold planes A / B only (chunk 0) are
    // issued: everything the first K-steps read.
#define SETUP_TILE(VB)                                                                                       \
    {                                                                                                        \
        int ptile_, ctile_;                                                                                  \
        if (MSK && a.lazy_order) {          /* position = id ((b T + t) n_ct + ct): sample-minor, one image per tile */ \
            const unsigned T_ = (unsigned)(a.N / a.in_mod), r_ = (unsigned)(VB) / (unsigned)n_ctiles;       \
            ctile_ = (int)((unsigned)(VB) - r_ * (unsigned)n_ctiles);                                        \
            const unsigned b_ = r_ / T_;                                                                     \
            ptile_ = (int)((r_ - b_ * T_) * (unsigned)a.in_mod + b_);                                        \
        } else {                                                                                             \
            xcd_tile_map((VB), n_ptiles, n_ctiles, ptile_, ctile_, a.xcd_split);                             \
        }                                                                                                    \
        ch0 = ctile_ * CT;                                                                                   \
        n0 = ptile_ * IMGS;                                                                                  \
        const _Float16* w0_ = ch0 < split ? a.wgt + (size_t)ch0 * Ktot : a.wgt_b + (size_t)(ch0 - split) * Ktot; \
        const _Float16* w1_ = ch0 + 128 < split ? a.wgt + (size_t)(ch0 + 128) * Ktot : a.wgt_b + (size_t)(ch0 + 128 - split) * Ktot; \
        rs_w0 = __builtin_amdgcn_make_buffer_rsrc((void*)w0_, 0, wbytes, 0x00020000);                        \
        if (!HALF) rs_w1 = __builtin_amdgcn_make_buffer_rsrc((void*)w1_, 0, wbytes, 0x00020000);             \
        if constexpr (IMAP) {                                                                                \
            rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, 2u * (unsigned)a.in_mod * HWC, 0x00020000); \
        } else {                                                                                             \
            const int nimg_ = a.N - n0 < IMGS ? a.N - n0 : IMGS;                                             \
            /* (one image per tile: a deterministic input of in_mod images serves every sample; launcher: in_mod >= N otherwise) */ \
            const int row0_ = IMGS == 1 ? n0 % a.in_mod : n0;                                                \
            rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)row0_ * HWC), 0, 2u * (unsigned)nimg_ * HWC, 0x00020000); \
            if constexpr (MSK)                                                                               \
                rs_bits = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in_bits + (size_t)n0 * (HWC >> 3)), 0, HWC >> 3, 0x00020000); \
        }                                                                                                    \
        ISSUE_W(0, 0);                                                                                       \
        if constexpr (IMAP) {                                                                                \
            tsel ^= 1;                                                                                       \
            if (tid < IMGS) row_tabs[tsel * 16 + tid] = n0 + tid < a.N ? a.imap[n0 + tid] : -1;              \
            __syncthreads();   /* (drains W(0) too: the dynamic-exit path only) */                           \
        }                                                                                                    \
        _Pragma("unroll") for (int k = 0; k < PRO; ++k) ISSUE_P(k, 0);                                       \
        ISSUE_BITS(0, 0, 1, 2, 3, 0, 0);                                                                     \
        ISSUE_BITS(1, 4, 9, -1, -1, 0, 0);                                                                   \
        ISSUE_W(s2_tap(1) * a.Cin, 1);                                                                       \
        if constexpr (NST == 4) ISSUE_W(s2_tap(2) * a.Cin, 2);                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    }

#define RAW_BARRIER()                                  \
    {                                                  \
        __builtin_amdgcn_sched_barrier(0);             \
        asm volatile("" ::: "memory");                 \
        __builtin_amdgcn_s_barrier();                  \
        asm volatile("" ::: "memory");                 \
        __builtin_amdgcn_sched_barrier(0);             \
    }
#define WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
    // What may still be in flight when K-step S of a chunk ends: the weight tile of the step after next (2 DMA instructions,
    // issued this step) and the patch pieces issued this step and the step before; everything older — in particular the next
    // step's weights and every piece issued two or more steps ago — has landed.  In the last chunk no next-chunk piece (steps
    // 5-8) and no weights beyond the tile's last step (steps 7, 8) are issued.
    // (Step 0 of a tile that follows a full tile, round 4: everything step 1 reads — weight stage 1, the A / B pieces — was issued BEFORE the
    //  previous tile's 16 output stores and has landed at the tile's start; only this step's own DMAs are younger than the stores, which may
    //  therefore stay in flight one step longer.)
#define END_OF_STEP_WAIT(S)                                                                                    \
    {                                                                                                          \
        if (S2_RELAX0 && (S) == 0 && chunk == 0 && stores16 && !IMAP && !S2_DRAIN_STORES) { WAIT_VM(NSTORES + G::wait_n(0, false)); } \
        else if (!last) { WAIT_VM(G::wait_n(S, false)); }                                                      \
        else { WAIT_VM(G::wait_n(S, true)); }                                                                  \
    }
    // One K-step = step S of the chunk's period (one tap x this chunk's 32 channels).  Two phases (LOAD part, barrier, MFMA part,
    // barrier); the two wave groups run one barrier apart.  Weight stage = S % 3 (9 steps per chunk: the index repeats).
    // Intervals between barriers, global step T: group 0 LOAD 4T+2k, MFMA 4T+2k+1; group 1 one later.
    //   WAR  weight stage (T+2)%3 was last read in 4T-3 (group 1, phase 0 of T-1; retired by its lgkmcnt(0) in 4T-2): refilled
    //        from 4T on.  A patch plane last read in step L (group 1's phase-1 reads retire in 4L+4) is refilled from step L+2.
    //   RAW  in interval 4T+3 every wave waits (counted vmcnt) until only the DMA of steps T and the pieces of T-1 are in flight.
#define S2_STEP(S)                                                                                             \
    {                                                                                                          \
        constexpr int pl_ = s2_plane(S), dy_ = s2_dy(S), dx_ = s2_dx(S), sn_ = (S) + NST - 1;                  \
        /* stage read by this step / filled for step S + NST - 1: compile-time with three stages (9 steps per chunk), */ \
        /* (chunk + S) & 3 with four (9 = 1 mod 4) */                                                          \
        const int st_r_ = NST == 3 ? (S) % 3 : ((chunk + (S)) & 3), st_w_ = NST == 3 ? ((S) + 2) % 3 : ((chunk + (S) + 3) & 3); \
        const char* ws_ = smem + G::wstage_off(st_r_) + a_off;                                                 \
        const char* pb_ = smem + (dy_ * G::cols(pl_) + dx_) * 64 + (boff[pl_] ^ (dy_ << 5));                   \
        /* MSK: pieces issued three steps ago have landed (this thread's wait at the end of the previous step).  The first due */ \
        /* piece is read BEFORE the fragment reads and finished behind them: its LDS round trip overlaps their issue — as one  */ \
        /* serial read-modify-write in front of them it stretched the LOAD part past the other group's MFMA part (2.03 -> 2.26 ms */ \
        /* on the 64 -> 128+128 launch of the headline) */                                                     \
        constexpr int due0_ = G::first_due(S);                                                                 \
        const bool due_on_ = MSK && due0_ >= 0 && (due0_ >= PRO ? true : (G::pstep(due0_ < 0 ? 0 : due0_) + 3 == (S) ? !last : chunk > 0)); \
        unsigned mw0_ = 0;                                                                                     \
        u32x4_m mv0_ = {0u, 0u, 0u, 0u};                                                                       \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                     \
            if constexpr (MSK && due0_ >= 0) { if (kk == S2_MASK_PHASE && due_on_) { MASK_LOAD(due0_ < 0 ? 0 : due0_, mw0_, mv0_); } } \
            if (kk == 0) {                                                                                     \
                _Pragma("unroll") for (int i = 0; i < TI; ++i) af[i] = *(const half8*)(ws_ + A_TILE(i));       \
            }                                                                                                  \
            _Pragma("unroll") for (int j = 0; j < JB; ++j)                                                     \
                if (!S2_ABL_HALFREADS || !(j & 1)) bf[j] = *(const half8*)(pb_ + G::cell_delta(pl_, JB * kk + j) * 64); \
                else bf[j] = bf[j - 1];                                                                        \
            if constexpr (MSK && due0_ >= 0) {                                                                 \
                if (kk == S2_MASK_PHASE) {                                                                     \
                    if (due_on_) { MASK_STORE(due0_ < 0 ? 0 : due0_, mw0_, mv0_); }                            \
                    _Pragma("unroll") for (int k = 0; k < NPT; ++k)                                            \
                        if (k != due0_ && MASK_DUE(S, k)) { if (due_on_) { MASK_P(k); } }                      \
                    /* (phase 1: the masked piece may be read in the next interval — the write has to be complete at the barrier) */ \
                    if (S2_MASK_PHASE == 1 && due_on_) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
                }                                                                                              \
            }                                                                                                  \
            if (kk == 0 || S2_WSPLIT) {                                                                        \
                /* weights of step S + NST - 1: of this chunk, or of the first steps of the next one (S2_WSPLIT: the two */ \
                /* DMA instructions of a wave in the two phases of the step) */                                \
                if (!S2_WSPLIT) {                                                                              \
                    if (sn_ <= 8) { ISSUE_W(s2_tap(sn_ <= 8 ? sn_ : 0) * a.Cin + c32, st_w_); }                \
                    else if (!last) { ISSUE_W(s2_tap(sn_ > 8 ? sn_ - 9 : 0) * a.Cin + c32 + 32, st_w_); }      \
                } else {                                                                                       \
                    if (sn_ <= 8) { ISSUE_W_HALF(s2_tap(sn_ <= 8 ? sn_ : 0) * a.Cin + c32, st_w_, kk); }       \
                    else if (!last) { ISSUE_W_HALF(s2_tap(sn_ > 8 ? sn_ - 9 : 0) * a.Cin + c32 + 32, st_w_, kk); } \
                }                                                                                              \
            }                                                                                                  \
            if (kk == S2_PIECE_PHASE) {                                                                        \
                _Pragma("unroll") for (int k = 0; k < NPT; ++k)                                                \
                    if (G::pstep(k) == (S)) {                                                                  \
                        if (k >= PRO) { ISSUE_P(k, c32); }                                                     \
                        else if (!last) { ISSUE_P(k, c32 + 32); }                                              \
                    }                                                                                          \
                if ((S) == 0) { ISSUE_BITS(2, 5, 6, 7, 8, c32, 0); }                                           \
                if ((S) == 4) { ISSUE_BITS(1, 4, 9, -1, -1, c32, 4); }                                         \
                if ((S) == 5) { if (!last) { ISSUE_BITS(0, 0, 1, 2, 3, c32 + 32, 0); } }                       \
            }                                                                                                  \
            /* (S2_MASK_PHASE 2: the masking of phase 0's MFMA part wrote LDS behind nothing but this phase's four fragment */ \
            /*  reads — LDS operations of a wave complete in order: with at most four outstanding the write is done) */ \
            if (MSK && S2_MASK_PHASE == 2 && kk == 1) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");       \
            if (kk == 1 && g == 1) END_OF_STEP_WAIT(S);        /* interval 4T+3, group 1: LOAD part */         \
            RAW_BARRIER();                                                                                     \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
            __builtin_amdgcn_s_setprio(1);                                                                     \
            if constexpr (MSK && S2_MASK_PHASE == 2 && due0_ >= 0) {                                           \
                if (kk == 0) {                                                                                 \
                    /* the piece's read-modify-write BETWEEN the MFMAs: its LDS round trip and its 20 vector instructions */ \
                    /* ride in the matrix pipe's shadow instead of stretching a LOAD part */                    \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[0][j] = mfma_16x16x32<BF>(af[0], bf[j], acc[0][j]); \
                    __builtin_amdgcn_sched_barrier(0);                                                         \
                    if (due_on_) { MASK_LOAD(due0_ < 0 ? 0 : due0_, mw0_, mv0_); }                             \
                    __builtin_amdgcn_sched_barrier(0);                                                         \
                    _Pragma("unroll") for (int i = 1; i < 3; ++i)                                              \
                        _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = mfma_16x16x32<BF>(af[i], bf[j], acc[i][j]); \
                    __builtin_amdgcn_sched_barrier(0);                                                         \
                    if (due_on_) { MASK_STORE(due0_ < 0 ? 0 : due0_, mw0_, mv0_); }                            \
                    _Pragma("unroll") for (int k = 0; k < NPT; ++k)                                            \
                        if (k != due0_ && MASK_DUE(S, k)) { if (due_on_) { MASK_P(k); } }                      \
                    __builtin_amdgcn_sched_barrier(0);                                                         \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[3][j] = mfma_16x16x32<BF>(af[3], bf[j], acc[3][j]); \
                } else {                                                                                       \
                    _Pragma("unroll") for (int i = 0; i < TI; ++i)                                             \
                        _Pragma("unroll") for (int j = 0; j < JB; ++j)                                         \
                            acc[i][JB * kk + j] = mfma_16x16x32<BF>(af[i], bf[j], acc[i][JB * kk + j]);        \
                }                                                                                              \
            } else {                                                                                           \
                _Pragma("unroll") for (int i = 0; i < TI; ++i)                                                 \
                    _Pragma("unroll") for (int j = 0; j < JB; ++j)                                             \
                        acc[i][JB * kk + j] = mfma_16x16x32<BF>(af[i], bf[j], acc[i][JB * kk + j]);            \
            }                                                                                                  \
            __builtin_amdgcn_s_setprio(0);                                                                     \
            if (kk == 1 && g == 0) END_OF_STEP_WAIT(S);        /* interval 4T+3, group 0: MFMA part */         \
            RAW_BARRIER();                                                                                     \
        }                                                                                                      \
    }

    // The walk: tile positions blockIdx.x, + gridDim.x, ... through xcd_tile_map — or, reading a deterministic input through keep bits
    // (MSK, ConvArgs::lazy_order), the sample-minor numbering id = (b T + t) n_ct + ct dealt per XCD: XCD x (workgroups = x mod 8, J of
    // them) owns the contiguous ids [x L, (x + 1) L) and its workgroup j takes x L + j, + J, ...: at any moment the XCD's workgroups read
    // ONE image's patch (consecutive samples of it), which therefore stays in the XCD's 4 MB L2 — with a run of ids per WORKGROUP (first
    // version) 32 different images of 131-512 KB each pass through it and every sample's read is a miss again.
    int vb = blockIdx.x, v_step = (int)gridDim.x, v_end = n_tiles;
    if (MSK && a.lazy_order) {
        if ((gridDim.x & 7) == 0) {
            const int L = (n_tiles + 7) >> 3, J = (int)gridDim.x >> 3;
            vb = (int)(blockIdx.x & 7) * L + (int)(blockIdx.x >> 3);
            v_step = J;
            v_end = (int)((blockIdx.x & 7) + 1) * L < n_tiles ? (int)((blockIdx.x & 7) + 1) * L : n_tiles;
        }                                                   // (a grid that is not a multiple of eight — fewer tiles than CUs: the ids in order)
        if (vb >= v_end) return;
    }
    SETUP_TILE(vb);
    constexpr int NSTORES = HALF ? 8 : 16;                 // output stores per thread of a full tile
    bool stores16 = false;                                 // the tile before this one issued exactly NSTORES stores per thread
    while (vb < v_end) {
        const int cur_ch0 = ch0, cur_n0 = n0, cur_tsel = tsel;
        accv acc[TI][TP];
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

        // weight stages 0, 1 and the A / B pieces of chunk 0 have landed.  The previous tile's 16 output stores per thread were
        // issued BEHIND them (vmcnt retires in order): a full tile leaves them in flight, a ragged one (some stores skipped:
        // the count is not known) and the dynamic-exit form drain everything.
        if (IMAP || !stores16 || S2_DRAIN_STORES) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORES) : "memory");
        if constexpr (MSK) {                               // the A / B pieces of chunk 0 (issued by SETUP_TILE)
#pragma unroll
            for (int k = 0; k < PRO; ++k) MASK_P(k);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        RAW_BARRIER();
        if (g == 1) RAW_BARRIER();                         // stagger
        half8 af[TI], bf[4];
        for (int chunk = 0; chunk < nC; ++chunk) {
            const bool last = chunk + 1 == nC;
            const int c32 = chunk * 32;
            S2_STEP(0) S2_STEP(1) S2_STEP(2) S2_STEP(3) S2_STEP(4) S2_STEP(5) S2_STEP(6) S2_STEP(7) S2_STEP(8)
        }
        if (g == 0) RAW_BARRIER();                         // both groups aligned: every wave is done with the stages and the patch
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        // ---- next tile's first weight stages and A / B pieces; they land during the epilogue below ----
        const int nvb = vb + v_step;
        if (nvb < v_end) SETUP_TILE(nvb);

        // ---- epilogue of the current tile: BN + ReLU on the accumulators, fp16 straight from the registers (S2_DIRECT) or through LDS, 32 KB per
        //      channel half, two rounds of 128 pixels (conv_igemm_wide_persist's, with this kernel's tile-pixel order) ----
        if (S2_ABL_NOEPI) {
            float sum_ = 0.f;
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) sum_ += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
            if (sum_ == 12345.678f) a.out[tid] = (_Float16)sum_;
        } else {
            char* const E = smem + G::E_OFF + g * 32768;
            int tl = tid & 255;
            asm volatile("" : "+v"(tl));                    // (per tile: hoisted out of the tile loop, what derives from it is spilled)
            const int chl = cur_ch0 + (HALF ? 0 : 128 * g);    // launch-wide channel of this group's channel 0 (BN table index)
            _Float16* outp = a.out;
            int oc = a.Cout, chg = chl;
            if (a.wgt_b) {
                if (chl >= split) { outp = a.out_b; oc = a.Cout - split; chg = chl - split; }
                else oc = split;
            }
            const int k = tl & 15;
            float* const poolp = TW == 4 ? ((a.wgt_b && chl >= split) ? a.pool_b : a.pool) : nullptr;     // wave-uniform (per channel half)
            if (poolp) {
                // ReLU + global average pool fused (the conv feeds an exit head only): a wave's pixel tile j is image wp * 8 + j of
                // the tile, its 16 pixels the 16 lanes of a DPP row — four v_add_f32 with DPP (quad xor 1, xor 2, half mirror, row
                // mirror) leave the sum in every lane; lane 0 of each row stores 4 consecutive channels as fp32.  Nothing goes
                // through LDS; the three barriers keep step with a channel half that takes the ordinary path.
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int cw = S2_DIRECT ? 32 * (i >> 1) + 8 * kq + 4 * (i & 1) : 16 * i + 4 * kq;      // this lane's 4 channels of tile i
                    const int c4 = chl + wc * 64 + cw;
                    const f32x4_e sc = *(const f32x4_e*)(bn_scale + c4), bi = *(const f32x4_e*)(bn_bias + c4);
#pragma unroll
                    for (int j = 0; j < TP; ++j) {
                        f32x4_e v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float x = fmaxf(__builtin_fmaf(acc[i][j][e], sc[e], bi[e]), 0.f);     // (explicitly fused: the same bits in every instantiation)
                            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));    // lane ^ 1
                            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));    // lane ^ 2
                            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));   // 7 - lane (half row)
                            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));   // 15 - lane (row)
                            v[e] = x * (1.f / 16.f);
                        }
                        int n = cur_n0 + wp * 8 + j;
                        if constexpr (IMAP) n = row_tabs[cur_tsel * 16 + wp * 8 + j];
                        else if (n >= a.N) n = -1;
                        if (l16 == 0 && n >= 0) *(f32x4_e*)(poolp + (size_t)n * oc + chg + wc * 64 + cw) = v;
                    }
                }
                if (!S2_DIRECT) { lds_barrier(); lds_barrier(); lds_barrier(); }
            } else if constexpr (S2_DIRECT) {
                // straight from the registers: lane (kq, l16) holds, for pixel tile j, channels 8 kq .. + 7 (tiles 0, 1) and 32 + 8 kq .. + 7
                // (tiles 2, 3) of the wave's 64 channels of tile pixel pbase + 16 j + l16; the four lanes of a pixel write 64 contiguous bytes
                const int cw = wc * 64 + 8 * kq;
                f32x4_e sc[4], bi[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    sc[i] = *(const f32x4_e*)(bn_scale + chl + cw + 32 * (i >> 1) + 4 * (i & 1));
                    bi[i] = *(const f32x4_e*)(bn_bias + chl + cw + 32 * (i >> 1) + 4 * (i & 1));
                }
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    const int p = pbase + 16 * j + l16;
                    int n = cur_n0 + G::p_img(p);
                    if constexpr (IMAP) n = row_tabs[cur_tsel * 16 + G::p_img(p)];      // tensor row, -1 beyond N
                    else if (n >= a.N) n = -1;
                    _Float16* const dst = outp + ((size_t)(n < 0 ? 0 : n) * (TW * TW) + G::p_oy(p) * TW + G::p_ox(p)) * oc + chg + cw;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        half8_e o;
#pragma unroll
                        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float v = acc[2 * h + ii][j][e] * sc[2 * h + ii][e] + bi[2 * h + ii][e];
                                if (a.relu) v = fmaxf(v, 0.f);
                                o[4 * ii + e] = a16_from_f32<BF>(v);
                            }
                        if (n >= 0 && !S2_ABL_NOSTORE) *(half8_e*)(dst + 32 * h) = o;
                    }
                }
            } else
#pragma unroll
            for (int rr = 0; rr < (HALF ? 1 : 2); ++rr) {   // (128-channel tiles: a group's 128 ch x 128 px are one round)
                if (rr) lds_barrier();                      // round 0's reads are done
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c4 = chl + wc * 64 + 16 * i + 4 * kq;
                    const f32x4_e sc = *(const f32x4_e*)(bn_scale + c4), bi = *(const f32x4_e*)(bn_bias + c4);
                    const int cq = wc * 8 + 2 * i + (kq >> 1);
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int p = wp * 64 + jj * 16 + l16;                   // pixel inside the round
                        half4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = acc[i][4 * rr + jj][e] * sc[e] + bi[e];
                            if (a.relu) v = fmaxf(v, 0.f);
                            o[e] = a16_from_f32<BF>(v);
                        }
                        *(half4*)(E + p * 256 + ((cq ^ l16) << 4) + (((kq ^ jj) & 1) << 3)) = o;
                    }
                }
                lds_barrier();
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {            // (two batches of four: the next tile's DMA sources stay in registers)
                    half8_e o8[4];
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int pl = (tl >> 4) + 16 * (4 * hb + it);
                        o8[it] = *(const half8_e*)(E + pl * 256 + ((k ^ (pl & 15)) << 4));
                    }
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int pl = (tl >> 4) + 16 * (4 * hb + it);
                        const int p = HALF ? g * 128 + pl : (pl >> 6) * 128 + rr * 64 + (pl & 63);     // tile pixel
                        int n = cur_n0 + G::p_img(p);
                        if constexpr (IMAP) n = row_tabs[cur_tsel * 16 + G::p_img(p)];      // tensor row, -1 beyond N
                        else if (n >= a.N) n = -1;
                        if (n < 0 || S2_ABL_NOSTORE) continue;
                        half8_e v = o8[it];
                        if (it & 1) v = __builtin_shufflevector(v, v, 4, 5, 6, 7, 0, 1, 2, 3);
                        _Float16* dst_ = outp + ((size_t)n * (TW * TW) + G::p_oy(p) * TW + G::p_ox(p)) * oc + chg + 8 * k;
                        if (S2_STORE_SC1) {
                            typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
                            // (hipcc does not model an asm store: without the trailing s_nop its next instruction may overwrite the data
                            // registers before the store has read them — cdna_hip_programming.md §5.7 item 1; the first build without it
                            // failed every model-level parity test)
                            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst_), "v"(__builtin_bit_cast(u32x4_, v)) : "memory");
                        } else {
                            *(half8_e*)dst_ = v;
                        }
                    }
                }
            }
        }
        stores16 = cur_n0 + IMGS <= a.N;              // (a pooled half issues 32 stores: vmcnt(16) then waits for half of them, never for less)
        vb = nvb;
    }
#undef S2_STEP
#undef END_OF_STEP_WAIT
#undef WAIT_VM
#undef RAW_BARRIER
#undef SETUP_TILE
#undef ISSUE_BITS
#undef MASK_DUE
#undef MASK_P
#undef MASK_STORE
#undef MASK_LOAD
#undef ISSUE_P
#undef ISSUE_W
#undef ISSUE_W_HALF
#undef A_TILE
}


// Shapes this kernel takes: 3x3 / stride 2 / pad 1, 32x32 -> 16x16, 16x16 -> 8x8 or 8x8 -> 4x4, Cout % 256 == 0 (both convs of
// a pair together), BN + ReLU epilogue.
bool conv_takes_s2_kernel(int ksize, int stride, int pad, int cin, int cout, int h, int w, int ho, int wo) {
    // (Cout % 256 != 0: the 128-channel tiles of the 16x16 output maps)
    return ksize == 3 && stride == 2 && pad == 1 && cin % 32 == 0 && (cout % 256 == 0 || (cout % 128 == 0 && ho == 16)) &&
           cout <= (ho == 4 ? 1024 : 512) && ho == wo && (ho == 16 || ho == 8 || ho == 4) && h == 2 * ho && w == 2 * wo;
}

template <int TW>
static int launch_s2(const ConvArgs& a, int n_cu, hipStream_t s) {
    using G = S2Geom<TW>;
    const bool half = a.Cout % 256 != 0;      // 128-channel tiles (TW == 16)
    const long tiles = (long)((a.N + G::IMGS - 1) / G::IMGS) * (a.Cout / (half ? 128 : 256));
    if (tiles <= 0 || tiles > 0x7fffffffL) return BMI_ERR_INVALID;
    const dim3 grid((unsigned)(tiles < n_cu ? tiles : n_cu)), block(512);
    if (half) {
        if constexpr (TW == 16) {
            if (a.wgt_b || (a.imap && a.in_bits)) return BMI_ERR_UNSUPPORTED;
            ConvArgs al = a;
            al.lazy_order = a.in_bits && opt_lazy_order() && a.in_mod < a.N && a.N % a.in_mod == 0;
            if (a.imap) {      // dynamic early exit: the same kernel family as the full run (the same bits for the images that go on)
                if (a.bf16) hipLaunchKernelGGL((conv3x3_s2_kernel<16, true, true, false, 128>), grid, block, 0, s, al, (int)tiles);
                else hipLaunchKernelGGL((conv3x3_s2_kernel<16, false, true, false, 128>), grid, block, 0, s, al, (int)tiles);
            } else if (a.in_bits) {
                if (a.bf16) hipLaunchKernelGGL((conv3x3_s2_kernel<16, true, false, true, 128>), grid, block, 0, s, al, (int)tiles);
                else hipLaunchKernelGGL((conv3x3_s2_kernel<16, false, false, true, 128>), grid, block, 0, s, al, (int)tiles);
            } else {
                if (a.bf16) hipLaunchKernelGGL((conv3x3_s2_kernel<16, true, false, false, 128>), grid, block, 0, s, al, (int)tiles);
                else hipLaunchKernelGGL((conv3x3_s2_kernel<16, false, false, false, 128>), grid, block, 0, s, al, (int)tiles);
            }
            BMI_CHECK_LAUNCH();
            return BMI_OK;
        } else {
            return BMI_ERR_UNSUPPORTED;
        }
    }
#define S2_LAUNCH(BF_, IMAP_) hipLaunchKernelGGL((conv3x3_s2_kernel<TW, BF_, IMAP_>), grid, block, 0, s, a, (int)tiles)
    if (a.in_bits) {
        if constexpr (TW == 16) {
            ConvArgs al = a;
            al.lazy_order = opt_lazy_order() && a.in_mod < a.N && a.N % a.in_mod == 0;
            if (a.bf16) hipLaunchKernelGGL((conv3x3_s2_kernel<16, true, false, true>), grid, block, 0, s, al, (int)tiles);
            else hipLaunchKernelGGL((conv3x3_s2_kernel<16, false, false, true>), grid, block, 0, s, al, (int)tiles);
        } else {
            return BMI_ERR_UNSUPPORTED;
        }
    }
    else if (a.imap) { if (a.bf16) S2_LAUNCH(true, true); else S2_LAUNCH(false, true); }
    else { if (a.bf16) S2_LAUNCH(true, false); else S2_LAUNCH(false, false); }
#undef S2_LAUNCH
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}

// BMI_ERR_UNSUPPORTED -> the caller goes on to conv_igemm_wide / conv_igemm.  The minimum-grid rule looks at the engine's
// FULL-CHUNK image count (ConvArgs::n_ref), never at the samples of this launch, like conv3x3_pw's: a t-shard runs the same
// kernel as the single-rank run and gets the same bits ("conv_s2" = 2 drops the rule: tests).
int launch_conv3x3_s2(const ConvArgs& a_in, hipStream_t s) {
    if (!opt_conv_s2() || a_in.in2 || a_in.partial || !conv_epilogue_is_plain(a_in)) return BMI_ERR_UNSUPPORTED;
    // keep bits on the input (ConvArgs::in_bits): the 32x32 -> 16x16 class (one image per tile), a deterministic, pre-scaled input
    if (a_in.in_bits && (a_in.Ho != 16 || a_in.imap || a_in.out_mul != 1.f || a_in.Cin % 32 != 0)) return BMI_ERR_UNSUPPORTED;
    if (a_in.lazy_planar && (!a_in.in_bits || (a_in.W & 1))) return BMI_ERR_UNSUPPORTED;
    if (!conv_takes_s2_kernel(a_in.ksize, a_in.stride, a_in.pad, a_in.Cin, a_in.Cout, a_in.H, a_in.W, a_in.Ho, a_in.Wo)) return BMI_ERR_UNSUPPORTED;
    ConvArgs a = a_in;
    if (a.N <= 0 || a.in_mod <= 0 || a.B <= 0) return BMI_ERR_INVALID;
    if (a.wgt_b && (!a.out_b || a.split <= 0 || a.split >= a.Cout || a.split % 128 != 0)) return BMI_ERR_INVALID;
    if ((a.pool || a.pool_b) && a.Ho != 4) return BMI_ERR_UNSUPPORTED;      // the pooled epilogue sums the 16 lanes of a DPP row = a 4x4 map
    // 32-bit byte offsets inside a buffer descriptor: a tile's own images (always), the whole tensor for the dynamic-exit form
    if (a.imap && (size_t)a.in_mod * a.H * a.W * a.Cin >= 0x7fffffffull) return BMI_ERR_UNSUPPORTED;
    if (a.in_mod < a.N && !a.imap && a.Ho != 16) return BMI_ERR_UNSUPPORTED;      // (tiles of several images take consecutive tensor rows)
    if ((size_t)a.H * a.W * a.Cin >= (1u << 24)) return BMI_ERR_UNSUPPORTED;
    static const int n_cu = [] {
        int dev = 0, cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cu = 0;
        return cu > 0 ? cu : 256;
    }();
    if (opt_conv_s2() != 2) {
        const int imgs = 256 / (a.Ho * a.Wo), n_sel = a.n_ref > 0 ? a.n_ref : a.N;
        if ((long)((n_sel + imgs - 1) / imgs) * (a.Cout / (a.Cout % 256 ? 128 : 256)) < 3 * n_cu / 4) return BMI_ERR_UNSUPPORTED;
    }
    a.xcd_split = xcd_split_for(a.Cout / (a.Cout % 256 ? 128 : 256), (size_t)a.Cout * 9 * a.Cin * 2);
    return a.Ho == 16 ? launch_s2<16>(a, n_cu, s) : (a.Ho == 8 ? launch_s2<8>(a, n_cu, s) : launch_s2<4>(a, n_cu, s));
}
