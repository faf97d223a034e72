// pt_ingest.hip — scene ingestion kernels: Assets/Resources/MeshProcessing.compute (ProcessMesh) and
// Assets/Resources/CopyTextureData.compute, one lane per triangle / texel as in the reference.  Both are pure
// gather-transform-scatter: 36-150 B read and 176 B written per triangle, 16 B read and 4 B written per texel (HBM-bound).
#include "pt_device.h"
#include "pt_launch.h"

namespace {

PT_DEV uint32_t load_u32(const uint8_t* base, uint32_t byteOffset) { return *(const uint32_t*)(base + byteOffset); }   // ByteAddressBuffer.Load
PT_DEV v3 load_f3(const uint8_t* base, uint32_t byteOffset)
{
    return mk3(pt_asfloat(load_u32(base, byteOffset)), pt_asfloat(load_u32(base, byteOffset + 4u)), pt_asfloat(load_u32(base, byteOffset + 8u)));
}
// mul(M, float4(v, w)).xyz with M in Unity memory order (element (r, c) at [c*4 + r])
PT_DEV v3 mul_point(const float* m, v3 v, float w)
{
    return mk3(m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12] * w,
               m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13] * w,
               m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14] * w);
}
// mul(float4(v, 0), M).xyz: row vector times matrix
PT_DEV v3 mul_row(v3 v, const float* m)
{
    return mk3(v.x * m[0] + v.y * m[1] + v.z * m[2] + 0.0f * m[3],
               v.x * m[4] + v.y * m[5] + v.z * m[6] + 0.0f * m[7],
               v.x * m[8] + v.y * m[9] + v.z * m[10] + 0.0f * m[11]);
}

__global__ __launch_bounds__(64) void pt_process_mesh(PTMeshDesc M, const uint8_t* __restrict__ vb, const uint8_t* __restrict__ ib,
                                                      float4* __restrict__ vertexPositionBuffer, float4* __restrict__ triangleAttributes)
{
    const uint32_t triIndex = blockIdx.x * 64u + threadIdx.x;
    if (triIndex >= M.TriangleCount) return;
    uint32_t i0, i1, i2;
    if (ib) {                                                           // HAS_INDEX_BUFFER
        if (M.flags & PT_MESH_HAS_32_BIT_INDICES) {
            const uint32_t byteOffset = triIndex * 12u;
            i0 = load_u32(ib, byteOffset); i1 = load_u32(ib, byteOffset + 4u); i2 = load_u32(ib, byteOffset + 8u);
        } else {                                                        // three 16-bit indices out of two aligned words (:76-85)
            const uint32_t byteOffset = triIndex * 6u;
            const uint32_t alignedOffset = byteOffset & ~3u;
            const uint32_t w0 = load_u32(ib, alignedOffset), w1 = load_u32(ib, alignedOffset + 4u);
            const bool even = (byteOffset % 4u) == 0u;
            i0 = even ? (w0 & 0xFFFFu) : (w0 >> 16);
            i1 = even ? (w0 >> 16) : (w1 & 0xFFFFu);
            i2 = even ? (w1 & 0xFFFFu) : (w1 >> 16);
        }
    } else { i0 = triIndex * 3u; i1 = i0 + 1u; i2 = i0 + 2u; }

    const uint32_t idx[3] = {i0, i1, i2};
    const uint32_t writeTriIndex = M.OutputTriangleStart + triIndex;
    v3 n[3] = {mk3(0.0f), mk3(0.0f), mk3(0.0f)}, t[3] = {mk3(0.0f), mk3(0.0f), mk3(0.0f)};
    float uvx[3] = {0.0f, 0.0f, 0.0f}, uvy[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint32_t base = idx[k] * M.VertexStride;
        const v3 p = mul_point(M.LocalToWorld, load_f3(vb, base + M.PositionOffset), 1.0f);
        vertexPositionBuffer[(size_t)writeTriIndex * 3u + k] = make_float4(p.x, p.y, p.z, 0.0f);
        if (M.flags & PT_MESH_HAS_NORMALS) n[k] = normalize3(mul_row(load_f3(vb, base + M.NormalOffset), M.WorldToLocal));   // inverse transpose
        if (M.flags & PT_MESH_HAS_TANGENTS) t[k] = normalize3(mul_point(M.LocalToWorld, load_f3(vb, base + M.TangentOffset), 0.0f));
        if (M.flags & PT_MESH_HAS_UVS) {
            uvx[k] = pt_asfloat(load_u32(vb, base + M.UVOffset));
            uvy[k] = pt_asfloat(load_u32(vb, base + M.UVOffset + 4u));
        }
    }
    float4* a = triangleAttributes + (size_t)writeTriIndex * 8u;       // PTTriangleAttributes = 8 x float4 (util/triangle_attributes.hlsl)
    a[0] = make_float4(n[0].x, n[0].y, n[0].z, 0.0f);
    a[1] = make_float4(n[1].x, n[1].y, n[1].z, 0.0f);
    a[2] = make_float4(n[2].x, n[2].y, n[2].z, 0.0f);
    a[3] = make_float4(t[0].x, t[0].y, t[0].z, 0.0f);
    a[4] = make_float4(t[1].x, t[1].y, t[1].z, 0.0f);
    a[5] = make_float4(t[2].x, t[2].y, t[2].z, 0.0f);
    a[6] = make_float4(uvx[0], uvy[0], uvx[1], uvy[1]);
    a[7] = make_float4(uvx[2], uvy[2], pt_asfloat(M.MaterialIndex), 0.0f);
}

__global__ __launch_bounds__(64) void pt_copy_texture_data(const float4* __restrict__ texture, uint32_t width, uint32_t height,
                                                           uint32_t dataOffset, uint32_t descriptorOffset, int hasAlpha, uint32_t* __restrict__ textureData)
{
    const uint32_t pixelX = blockIdx.x * 8u + (threadIdx.x & 7u), pixelY = blockIdx.y * 8u + (threadIdx.x >> 3);   // [numthreads(8, 8, 1)]
    if (pixelX >= width || pixelY >= height) return;
    if (pixelX == 0u && pixelY == 0u) {       // every reference thread writes the same four words; one lane is enough
        textureData[descriptorOffset + 0u] = width;
        textureData[descriptorOffset + 1u] = height;
        textureData[descriptorOffset + 2u] = dataOffset;
        textureData[descriptorOffset + 3u] = 0u;
    }
    const float4 texel = texture[(size_t)pixelY * width + pixelX];
    const uint32_t r = pt_f2u(texel.x * 255.0f) & 0xFFu, g = pt_f2u(texel.y * 255.0f) & 0xFFu, b = pt_f2u(texel.z * 255.0f) & 0xFFu;
    const uint32_t a = hasAlpha ? (pt_f2u(texel.w * 255.0f) & 0xFFu) : 0xFFu;
    textureData[(size_t)pixelY * width + pixelX + dataOffset] = (a << 24) | (b << 16) | (g << 8) | r;
}

} // namespace

hipError_t pt_launch_process_mesh(const PTMeshDesc& M, const void* dVertexBuffer, const void* dIndexBuffer, float4* dVertexPositions,
                                  float4* dTriangleAttributes, hipStream_t stream)
{
    if (M.TriangleCount == 0u) return hipSuccess;
    hipLaunchKernelGGL(pt_process_mesh, dim3((M.TriangleCount + 63u) / 64u), dim3(64), 0, stream, M, (const uint8_t*)dVertexBuffer,
                       (const uint8_t*)dIndexBuffer, dVertexPositions, dTriangleAttributes);
    return hipGetLastError();
}

hipError_t pt_launch_copy_texture(const float4* dTexture, uint32_t width, uint32_t height, uint32_t dataOffset, uint32_t descriptorOffset,
                                  int hasAlpha, uint32_t* dTextureData, hipStream_t stream)
{
    hipLaunchKernelGGL(pt_copy_texture_data, dim3((width + 7u) / 8u, (height + 7u) / 8u), dim3(64), 0, stream, dTexture, width, height,
                       dataOffset, descriptorOffset, hasAlpha, dTextureData);
    return hipGetLastError();
}
