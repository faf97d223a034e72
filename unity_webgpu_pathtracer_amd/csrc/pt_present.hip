// pt_present.hip — the presentation blit: Assets/Resources/Presentation.shader:36-73 with util/tonemap.hlsl as one
// elementwise kernel (16 B read + 16 B written per pixel: HBM-bound, one wave per 64 consecutive pixels, coalesced).
// The reference runs it as a full-screen fragment shader over _outputRT[_currentRT] (PathTracer.cs:255-266); source and
// destination have the same size there, so tex2D at the pixel centre is the pixel itself.
#include "pt_device.h"
#include "pt_launch.h"

namespace {

PT_DEV v3 pow3(v3 c, float e) { return mk3(pt_pow(c.x, e), pt_pow(c.y, e), pt_pow(c.z, e)); }
PT_DEV v3 pow33(v3 c, v3 e) { return mk3(pt_pow(c.x, e.x), pt_pow(c.y, e.y), pt_pow(c.z, e.z)); }
PT_DEV v3 div33(v3 a, v3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
PT_DEV v3 lerp33(v3 a, v3 b, v3 t) { return a + t * (b - a); }                 // HLSL lerp: x + s (y - x)
PT_DEV v3 mul33(const float (&m)[3][3], v3 c)
{
    return mk3(m[0][0] * c.x + m[0][1] * c.y + m[0][2] * c.z,
               m[1][0] * c.x + m[1][1] * c.y + m[1][2] * c.z,
               m[2][0] * c.x + m[2][1] * c.y + m[2][2] * c.z);
}

// util/tonemap.hlsl:6-11
PT_DEV v3 linear_to_srgb(v3 rgb)
{
    v3 low = rgb * 12.92f;
    v3 high = pow3(rgb, 1.0f / 2.4f) * 1.055f - mk3(0.055f);
    v3 gt = mk3(rgb.x > 0.0031308f ? 1.0f : 0.0f, rgb.y > 0.0031308f ? 1.0f : 0.0f, rgb.z > 0.0031308f ? 1.0f : 0.0f);
    return lerp33(low, high, gt);
}

// :21-46
PT_DEV v3 tm_aces(v3 color)
{
    const float in[3][3] = {{0.59719f, 0.35458f, 0.04823f}, {0.07600f, 0.90834f, 0.01566f}, {0.02840f, 0.13383f, 0.83777f}};
    const float out[3][3] = {{1.60475f, -0.53108f, -0.07367f}, {-0.10208f, 1.10813f, -0.00605f}, {-0.00327f, -0.07276f, 1.07602f}};
    color = mul33(in, color);
    v3 a = color * (color + mk3(0.0245786f)) - mk3(0.000090537f);
    v3 b = color * (0.983729f * color + mk3(0.4329510f)) + mk3(0.238081f);
    color = div33(a, b);
    return mul33(out, color);
}

// :49-54
PT_DEV v3 tm_filmic(v3 x)
{
    v3 X = mk3(pt_max(0.0f, x.x - 0.004f), pt_max(0.0f, x.y - 0.004f), pt_max(0.0f, x.z - 0.004f));
    v3 result = div33(X * (6.2f * X + mk3(0.5f)), X * (6.2f * X + mk3(1.7f)) + mk3(0.06f));
    return pow3(result, 2.2f);
}

// :57-74.  b and c depend on constants only: the launcher evaluates them once on the host with the same canonical pt_pow
// (bit-identical to evaluating them per pixel) instead of five pow() per pixel here.
struct LottesConsts { float a, ad, b, c; };
PT_HD LottesConsts lottes_consts()
{
    const float a = 1.6f, d = 0.977f, hdrMax = 8.0f, midIn = 0.18f, midOut = 0.267f;
    LottesConsts k;
    k.a = a;
    k.ad = a * d;
    k.b = (-pt_pow(midIn, a) + pt_pow(hdrMax, a) * midOut) / ((pt_pow(hdrMax, a * d) - pt_pow(midIn, a * d)) * midOut);
    k.c = (pt_pow(hdrMax, a * d) * pt_pow(midIn, a) - pt_pow(hdrMax, a) * pt_pow(midIn, a * d) * midOut) /
          ((pt_pow(hdrMax, a * d) - pt_pow(midIn, a * d)) * midOut);
    return k;
}
PT_DEV v3 tm_lottes(v3 x, const LottesConsts& k) { return div33(pow3(x, k.a), pow3(x, k.ad) * k.b + mk3(k.c)); }

// :76-79
PT_DEV v3 tm_reinhard(v3 x) { return div33(x, mk3(1.0f) + x); }

__global__ __launch_bounds__(256) void pt_present(PTPresentParams Q, LottesConsts LK, const float4* __restrict__ src, float4* __restrict__ dst)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t n = Q.OutputWidth * Q.OutputHeight;
    if (i >= n) return;
    const uint32_t px = i % Q.OutputWidth, py = i / Q.OutputWidth;
    const float4 s = src[i];
    v3 color = mk3(s.x, s.y, s.z);
    color = color * Q.Exposure;
    switch (Q.Mode) {
    case 1: color = tm_aces(color); break;
    case 2: color = tm_filmic(color); break;
    case 3: color = tm_reinhard(color); break;
    case 4: color = tm_lottes(color, LK); break;
    default: break;
    }
    if (Q.sRGB) color = linear_to_srgb(color);
    // contrast and clamp, brightness, saturation (Presentation.shader:61-68)
    color = mk3(0.5f) + Q.Contrast * (color - mk3(0.5f));
    color = mk3(pt_saturate(color.x), pt_saturate(color.y), pt_saturate(color.z));
    color = pow3(color, 1.0f / Q.Brightness);
    const float l = luminance3(color);
    color = mk3(l) + Q.Saturation * (color - mk3(l));
    // vignette (:70-71); i.uv is the pixel centre
    const float cu = (((float)px + 0.5f) / (float)Q.OutputWidth - 0.5f) * 2.0f;
    const float cv = (((float)py + 0.5f) / (float)Q.OutputHeight - 0.5f) * 2.0f;
    color = color * (1.0f - (cu * cu + cv * cv) * Q.Vignette);
    dst[i] = make_float4(color.x, color.y, color.z, 1.0f);
}

} // namespace

hipError_t pt_launch_present(const PTPresentParams& Q, const float4* src, float4* dst, hipStream_t stream)
{
    const uint32_t n = Q.OutputWidth * Q.OutputHeight;
    hipLaunchKernelGGL(pt_present, dim3((n + 255u) / 256u), dim3(256), 0, stream, Q, lottes_consts(), src, dst);
    return hipGetLastError();
}
