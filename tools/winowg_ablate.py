"""Developer tool: print a timing-only variant of csrc/winowg.hip (see tools/winowg_ablation.sh).  usage: winowg_ablate.py <variant>"""
import os
import sys

src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zsgnet-pytorch_amd", "csrc", "winowg.hip")).read()
v = sys.argv[1]


def sub(old, new, count=1):
    global src
    assert src.count(old) >= 1, old
    src = src.replace(old, new) if count == 0 else src.replace(old, new, count)


if v == "noload":          # no global loads: the staged registers are loop-invariant values
    sub("rd[a * 2 + bb] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_a, (int)(ok ? dyv[a * 2 + bb] : ZSG_OOB), so_d, 0));",
        "rd[a * 2 + bb] = f32x4{(float)so_d, 1.f, 2.f, (float)ok};")
    sub("rx[col] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_b, (int)(ok ? xv[col] : ZSG_OOB), so_x, 0));",
        "rx[col] = f32x4{(float)so_x, 1.f, 2.f, (float)ok};")
elif v == "notransform":   # loads and LDS stores stay, the transform arithmetic goes
    sub("r0[e] = fmaf(beta, rd[2][e], rd[0][e]);", "r0[e] = rd[0][e];")
    sub("r1[e] = fmaf(beta, rd[3][e], rd[1][e]);", "r1[e] = rd[1][e];")
    sub("*(f32x4*)(ps + 4 * SP) = r0 + r1;", "*(f32x4*)(ps + 4 * SP) = rd[2];")
    sub("*(f32x4*)(ps + 8 * SP) = r0 - r1;", "*(f32x4*)(ps + 8 * SP) = rd[3];")
    sub("*(f32x4*)(ps + 12 * SP) = -r1;", "*(f32x4*)(ps + 12 * SP) = r1;")
    sub("t[0] = rx[0] - rx[2];", "t[0] = rx[0];")
    sub("t[1] = rx[1] + rx[2];", "t[1] = rx[1];")
    sub("t[2] = rx[2] - rx[1];", "t[2] = rx[2];")
    sub("t[3] = rx[1] - rx[3];", "t[3] = rx[3];")
    sub("v[e] = fmaf(ww_quad_other(t[j][e]), sgn, t[j][e]);", "v[e] = t[j][e];")
elif v == "nomfma":
    sub("acc[pl] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[po + ks * 128], b[po + ks * 128], acc[pl], 0, 0, 0);",
        "{ acc[pl][ks] += a[po + ks * 128] * b[po + ks * 128]; }")
elif v == "nofrag":        # MFMAs on registers that never come from LDS
    sub("acc[pl] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[po + ks * 128], b[po + ks * 128], acc[pl], 0, 0, 0);",
        "{ float fa_ = (float)(po + ks), fb_ = (float)(pl - ks); asm volatile(\"\" : \"+v\"(fa_), \"+v\"(fb_)); acc[pl] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_, fb_, acc[pl], 0, 0, 0); }")
elif v == "noepilogue":    # stage loop only: no output transform, no slab / dw stores (one element keeps the accumulators alive)
    sub("    if (n_st <= 0) return;\n", "    if (n_st <= 0) return;\n    { float s_ = 0.f; for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) s_ += acc[i][e]; if (s_ == 12345.678f) p.dw[0] = s_; return; }\n")
elif v == "nobarrier":     # the stage's closing barrier removed (races: timing only)
    sub("        store_stage((it + 1) & 1);\n        __syncthreads();", "        store_stage((it + 1) & 1);")
else:
    assert v == "full", v
print(src)
