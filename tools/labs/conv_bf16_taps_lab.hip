// LAB (round 6, measured and NOT shipped): the flat-position 3 x 3 bf16 convolution with fat waves (a wave owns 128 positions x 64 channels).  This is the kernel
// text as it ran inside fcn8s_tensorflow_amd/csrc/gemm_bf16.hip (it uses that file's Bf16Conv256Args, glds16b, wait_vm, xcd_run and conv_rows_epilogue and is not
// built on its own); tests/test_ops_gpu.py held it bit-identical to conv_bf16_rows_kernel<64>.  Results at 4 x 2048x1024 (profiles/r06_bf16_fat_tile_lab.txt):
// conv4_2 forward 0.633 ms against 0.55-0.56 for the shipped 32 x 64 wave tile, conv3_2 0.72 against 0.57, conv2_2 0.99 against 0.64, conv5_x 0.235 against 0.165.
// An earlier form with eight waves in one 147 KB block per CU (256 x 256 and 512 x 128 blocks, K-tile = filter row) was slower still: conv4_2 0.73 ms.
// Round 6: the same flat-position product with FAT waves, two small blocks per CU.  A wave of conv_bf16_rows_kernel owns 32 positions x 64 columns: 18 fragment
// reads for 12 MFMAs per K-tile (1.5 LDS reads per MFMA -- the LDS 75 % as busy as the matrix pipe would be at full rate) and a block barrier every 384
// MFMA-cycles; rocprofv3 had its waves parked at that barrier for 0.30-0.50 of their cycles and the matrix pipes 0.29-0.53 busy
// (profiles/r05_c5_bf16_train_wave_state.txt).  Here a block is FOUR waves (2 x 2) on 256 positions x 128 columns, a wave owns 128 x 64 = 4 x 2 accumulators
// (128 VGPRs): 12 reads for 16 MFMAs per tap (0.75), 213 bytes of LDS-DMA per MFMA instead of 302.  Two blocks per CU (73 KB of LDS each) put two waves of
// DIFFERENT blocks on every SIMD: one block's barrier, DMA wait and first fragment reads run under the other's MFMAs.  [The first attempt kept eight waves in one
// 147 KB block per CU, K-tile = filter row as above: 30 % SLOWER than the 32 x 64 form -- lock-step waves serialise DMA wait, reads and MFMAs, and the 48 KB B image
// of a K-tile, issued one tile ahead, did not land in time: with its LDS-DMA off conv4_2's forward pass took 0.56 ms, without reads 0.69, without MFMAs 0.43,
// complete 0.73 (profiles/r06_bf16_fat_tile_lab.txt).]
//   step s = (K-tile kt = (channel chunk, filter row), tap tx): the A image of a K-tile (272 rows x 64 B, two stages, issued one K-tile = three steps ahead) is
//   shared by its three steps; the B slice of a step (128 columns x 64 B = 8 KB) lives in a ring of five, issued FOUR steps ahead into the slot the previous step
//   has just finished reading.  One barrier per step.  Per wave and step: 8 + 4 ds_read_b128, 16 MFMAs; the second half's fragments are read under the first
//   half's MFMAs.
__global__ __launch_bounds__(256, 2) void conv_bf16_taps_kernel(const Bf16Conv256Args p)
{
    constexpr int BN = 128, BM = 256, TM = 4, AROWS = BM + 16, NAC = AROWS / 16, NBC = BN / 16;       // 17 A chunks, 8 B chunks of 16 rows (1 KB)
    constexpr int ABYTES = AROWS * G_ROWB, BBYTES = BN * G_ROWB, NSA = 2, NSB = 5, BOFF = NSA * ABYTES;
    constexpr int NAI = (NAC + 3) / 4;
    static_assert(NSA * ABYTES + NSB * BBYTES <= 80 * 1024, "two blocks per CU");
    __shared__ __attribute__((aligned(16))) unsigned char smem[NSA * ABYTES + NSB * BBYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wn = wave & 1;
    const int Wp = p.W + 2;
    const unsigned ntn = (unsigned)(p.Cout / BN);
    const unsigned lid = xcd_run(blockIdx.x, gridDim.x);
    const unsigned tmi = lid / ntn, tni = lid % ntn;               // column tiles of one row tile are neighbours: they share its A rows behind one L2
    const long long q0 = (long long)tmi * BM; const int n0 = (int)tni * BN;
    unsigned a_voff[NAI], b_voff[2];
#pragma unroll
    for (int i = 0; i < NAI; ++i) {
        const int chunk = wave + 4 * i, row = chunk * 16 + lane / 4, pc = lane % 4, lc = pc ^ ((row >> 2) & 3);
        a_voff[i] = p.xp_ps ? (unsigned)((row * 32 + lc * 8) * 2) : (unsigned)(((long long)row * p.Cin + lc * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave + 4 * i) * 16 + lane / 4, pc = lane % 4, lc = pc ^ ((row >> 2) & 3);
        b_voff[i] = W_PLANES ? (unsigned)((row * 32 + lc * 8) * 2) : (unsigned)(((long long)row * 9 * p.Cin + lc * 8) * 2);
    }
    const unsigned short* a_base = p.xp + (q0 - Wp - 1) * (p.xp_ps ? 32 : p.Cin);
    const unsigned short* b_base = W_PLANES ? p.wt + (long long)n0 * 32 : p.wt + (long long)n0 * 9 * p.Cin;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int nci = p.Cin / G_BK, nkt = 3 * nci, nsteps = 3 * nkt;
    const int na = wave == 0 ? 5 : 4;                              // this wave's A instructions per K-tile (chunks wave, wave + 4, ...)
    // the issue state walks forward with the steps: next A K-tile (channel chunk ia_c, filter row ia_ty) and next B step (chunk ib_c, filter row ib_ty, tap ib_tx)
    int ia_c = 0, ia_ty = 0, ia_st = 0, ib_c = 0, ib_ty = 0, ib_tx = 0, ib_slot = 0;
    auto issue_a = [&]() {
        const unsigned st = lds0 + (unsigned)(ia_st * ABYTES);
        const unsigned short* ga = p.xp_ps ? a_base + (long long)ia_c * p.xp_ps + (long long)ia_ty * Wp * 32 : a_base + (long long)ia_ty * Wp * p.Cin + ia_c * G_BK;
#pragma unroll
        for (int i = 0; i < NAI; ++i) if (wave + 4 * i < NAC) glds16b(ga, a_voff[i], st + (unsigned)((wave + 4 * i) * 1024));
        ia_st ^= 1;
        if (++ia_ty == 3) { ia_ty = 0; ++ia_c; }
    };
    auto issue_b = [&]() {
        const unsigned st = lds0 + (unsigned)(BOFF + ib_slot * BBYTES);
        // weight planes [k / 32][Cout][32] with k = (tap, ci): the slice of (tap t, chunk c) is plane t * nci + c
        const unsigned short* gb = W_PLANES ? b_base + (long long)((ib_ty * 3 + ib_tx) * nci + ib_c) * p.Cout * 32 : b_base + (long long)(ib_ty * 3 + ib_tx) * p.Cin + ib_c * G_BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16b(gb, b_voff[i], st + (unsigned)((wave + 4 * i) * 1024));
        ib_slot = ib_slot + 1 == NSB ? 0 : ib_slot + 1;
        if (++ib_tx == 3) { ib_tx = 0; if (++ib_ty == 3) { ib_ty = 0; ++ib_c; } }
    };
    f32x16 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int arow = wr * (TM * 32) + (lane & 31);
    // prologue, in the order the waits below count on: A(0), B(0), B(1), B(2), B(3)
    issue_a();
#pragma unroll
    for (int j = 0; j < 4; ++j) if (j < nsteps) issue_b();
    int rs = 0, ra = 0, tx = 0;                                    // ring slot of this step's B slice, stage of this K-tile's A image, tap of this step
    for (int s = 0; s < nsteps; ++s) {
        // What may still be in flight when step s starts (this wave's instructions, issue order = step order, inside a step B first, then A):
        //   tx = 0: B(s) [issued in step s - 4] and A(kt) [step s - 3] must have landed -> only steps s - 2, s - 1 may be outstanding: 2 + 2 B
        //   tx = 1: B(s) [step s - 4]; behind it: that step's A (landed: the step before waited for it), steps s - 3, s - 2: 2 + 2 B, step s - 1: 2 B + A(kt + 1)
        //   tx = 2: B(s) [step s - 4]; behind it: step s - 3: 2 B, step s - 2: 2 B + A(kt + 1), step s - 1: 2 B
        // (the last four steps issue nothing behind them: everything must have landed)
        if (s + 4 >= nsteps) wait_vm<0>();
        else if (tx == 0) wait_vm<4>();
        else if (na == 5) wait_vm<11>(); else wait_vm<10>();
        __builtin_amdgcn_s_barrier();                              // step s has landed everywhere; everybody is done reading step s - 1's B slot (and, at tx = 0, K-tile kt - 1's A stage)
        const unsigned char* st = smem + ra * ABYTES;
        const unsigned char* stb = smem + BOFF + rs * BBYTES;
        bf16x8 af[2][TM], bfr[2][2];
        auto frags = [&](int ks) {
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const int rb = wn * 64 + tn * 32 + (lane & 31), pb = (2 * ks + (lane >> 5)) ^ ((rb >> 2) & 3);
                bfr[ks][tn] = *reinterpret_cast<const bf16x8*>(stb + rb * G_ROWB + pb * 16);
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int r = arow + tm * 32 + tx, pc = (2 * ks + (lane >> 5)) ^ ((r >> 2) & 3);
                af[ks][tm] = *reinterpret_cast<const bf16x8*>(st + r * G_ROWB + pc * 16);
            }
        };
        auto mfmas = [&](int ks) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][tm], bfr[ks][tn], acc[tm][tn], 0, 0, 0);
        };
        frags(0);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 4 < nsteps) issue_b();                             // into the slot step s - 1 read
        if (tx == 0 && s + 3 < nsteps) issue_a();                  // the next K-tile's image into the stage K-tile kt - 1 read
        __builtin_amdgcn_sched_barrier(0);
        frags(1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(1);
        rs = rs + 1 == NSB ? 0 : rs + 1;
        if (++tx == 3) { tx = 0; ra ^= 1; }
    }
    conv_rows_epilogue<TM, 2, 2>(p, acc, smem, tid, q0, n0, tmi);
}

