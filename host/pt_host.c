/*
 * host/pt_host.c — plain-C host of the drop-in library (north_star: "host code stays in C calling HIP through
 * that same thin C-ABI").  It does what the Unity C# host does for the hot path, through include/ptmi_plugin.h only:
 *   BVHScene.cs:629-667   BuildBVH -> GetCWBVHNodesSize/TrisSize/Data -> upload          (here: PTSetScene)
 *   PathTracer.cs:226-272 per-frame uniforms -> dispatch -> _currentSample += spp -> flip (here: PTRenderPass/PTFlipFrames)
 * Scene: the Cornell box of SURVEY.md 8d (8 triangles, one rectangle light).
 *
 *   usage: pt_host [width height spp passes out.f32 [abi_dump.bin]]
 *          PT_HOST_DEVICES=0,1,2,3 (environment): render on those devices through PTCreateMulti / PTGroup* -- one process, N
 *          GPUs, frame assembled in the library from every device's tiles; the same device may be listed twice.
 *          out.f32: width*height float4, linear radiance, row 0 = bottom.
 *          abi_dump.bin: everything this host handed across the C-ABI, byte for byte -- the BuildBVH input, the attribute,
 *          material and light arrays, the CWBVH the library returned, and the PTFrameParams of every pass -- so that a
 *          checker can render the very same inputs (tests/test_gpu_parity.py::test_plain_c_host compares bit for bit).
 */
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ptmi_plugin.h"

#define LOAD(name) \
    __typeof__(&name) p_##name = (__typeof__(&name))dlsym(lib, #name); \
    if (!p_##name) { fprintf(stderr, "missing symbol %s\n", #name); return 2; }

static void quad(PTFloat4* v, PTTriangleAttributes* a, int q, const float* o, const float* eu, const float* ev,
                 const float* n, unsigned material)
{
    float c[4][3];
    for (int k = 0; k < 3; k++) {
        c[0][k] = o[k]; c[1][k] = o[k] + eu[k]; c[2][k] = o[k] + eu[k] + ev[k]; c[3][k] = o[k] + ev[k];
    }
    const int idx[2][3] = {{0, 1, 2}, {0, 2, 3}};
    const float uv[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
    float tl = sqrtf(eu[0] * eu[0] + eu[1] * eu[1] + eu[2] * eu[2]);
    for (int t = 0; t < 2; t++) {
        PTTriangleAttributes* at = &a[q * 2 + t];
        memset(at, 0, sizeof(*at));
        float* nn[3] = {at->normal0, at->normal1, at->normal2};
        float* tt[3] = {at->tangent0, at->tangent1, at->tangent2};
        float* uu[3] = {at->uv0, at->uv1, at->uv2};
        for (int k = 0; k < 3; k++) {
            PTFloat4* p = &v[(q * 2 + t) * 3 + k];
            p->x = c[idx[t][k]][0]; p->y = c[idx[t][k]][1]; p->z = c[idx[t][k]][2]; p->w = 0.0f;   /* MeshProcessing.compute:100-102 */
            for (int j = 0; j < 3; j++) { nn[k][j] = n[j]; tt[k][j] = eu[j] / tl; }
            uu[k][0] = uv[idx[t][k]][0]; uu[k][1] = uv[idx[t][k]][1];
        }
        at->materialIndex = material;
    }
}

static void material(PTMaterialData* m, float r, float g, float b)
{
    memset(m, 0, sizeof(*m));                                  /* BVHScene.cs:241-282 */
    m->baseColor[0] = powf(r, 2.2f); m->baseColor[1] = powf(g, 2.2f); m->baseColor[2] = powf(b, 2.2f);
    m->opacity = 1.0f; m->alphaCutoff = 0.5f; m->roughness = 1.0f; m->normalScale = 1.0f; m->ior = 1.1f;
    m->texBaseColor = m->texMetallicRoughness = m->texNormal = m->texEmission = m->texOcclusion = m->texPad = -1.0f;
    m->uvScale[0] = m->uvScale[1] = 1.0f;
}

int main(int argc, char** argv)
{
    int W = argc > 1 ? atoi(argv[1]) : 256, H = argc > 2 ? atoi(argv[2]) : 256;
    int spp = argc > 3 ? atoi(argv[3]) : 1, passes = argc > 4 ? atoi(argv[4]) : 1;
    const char* out = argc > 5 ? argv[5] : NULL;
    const char* dumpPath = argc > 6 ? argv[6] : NULL;
    FILE* dump = dumpPath ? fopen(dumpPath, "wb") : NULL;
    if (dumpPath && !dump) { fprintf(stderr, "cannot write %s\n", dumpPath); return 2; }
    /* the HOST's choice, before the first HIP call: passes overlap on separate streams, and streams that share one of the runtime's
     * hardware queues (default 4) serialise; the library sizes its passes in flight to this variable */
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    const char* path = getenv("PT_PLUGIN");
    if (!path) path = "unity_webgpu_pathtracer_amd/lib/libunity-webgpu-pathtracer-plugin.so";
    void* lib = dlopen(path, RTLD_NOW);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    LOAD(BuildBVH) LOAD(DestroyBVH) LOAD(IsBVHReady) LOAD(GetCWBVHNodesSize) LOAD(GetCWBVHTrisSize) LOAD(GetCWBVHData)
    LOAD(PTCreate) LOAD(PTDestroy) LOAD(PTSetScene) LOAD(PTRenderPass) LOAD(PTFlipFrames) LOAD(PTReadback)
    LOAD(PTGetStats) LOAD(PTGetLastError) LOAD(PTSynchronize)
    LOAD(PTCreateMulti) LOAD(PTGroupDestroy) LOAD(PTGroupSetScene) LOAD(PTGroupRenderPass) LOAD(PTGroupFlipFrames) LOAD(PTGroupReadback)
    LOAD(PTGroupGetStats)
    int devices[64], nDevices = 0;
    const char* devList = getenv("PT_HOST_DEVICES");
    if (devList && *devList) {
        char* copy = strdup(devList);
        for (char* tok = strtok(copy, ","); tok && nDevices < 64; tok = strtok(NULL, ",")) devices[nDevices++] = atoi(tok);
        free(copy);
    }

    PTFloat4 verts[24];
    PTTriangleAttributes attrs[8];
    const float o0[3] = {-1, 0, -1}, o1[3] = {-1, 0, 1}, o3[3] = {1, 0, -1};
    const float ex[3] = {2, 0, 0}, ey[3] = {0, 2, 0}, ez[3] = {0, 0, 2};
    const float ny[3] = {0, 1, 0}, nzm[3] = {0, 0, -1}, nx[3] = {1, 0, 0}, nxm[3] = {-1, 0, 0};
    quad(verts, attrs, 0, o0, ex, ez, ny, 0);      /* floor            */
    quad(verts, attrs, 1, o1, ex, ey, nzm, 0);     /* back wall z = +1 */
    quad(verts, attrs, 2, o0, ez, ey, nx, 1);      /* left, red        */
    quad(verts, attrs, 3, o3, ez, ey, nxm, 2);     /* right, green     */
    PTMaterialData mats[3];
    material(&mats[0], 0.7169812f, 0.7169812f, 0.7169812f);
    material(&mats[1], 1.0f, 0.0f, 0.0f);
    material(&mats[2], 0.14809364f, 0.9622642f, 0.13163051f);
    PTLight light;                                  /* PathTracer.cs:343-348, 442-460 */
    memset(&light, 0, sizeof(light));
    light.position[0] = -0.25f; light.position[1] = 1.98f; light.position[2] = -0.25f;
    light.type = PT_LIGHT_TYPE_RECTANGLE;
    light.emission[0] = 17; light.emission[1] = 12; light.emission[2] = 4; light.range = 100;
    light.u[0] = 0.5f; light.v[2] = 0.5f; light.area = 0.25f;

    int h = p_BuildBVH(verts, 8);
    if (h < 0 || !p_IsBVHReady(h)) { fprintf(stderr, "BuildBVH failed\n"); return 1; }
    PTFloat4 *nodes = NULL, *tris = NULL;
    p_GetCWBVHData(h, &nodes, &tris);
    if (dump) {
        /* header: magic, nodes bytes, tris bytes, triangles, materials, lights, passes, sizeof(PTFrameParams) */
        const uint32_t hdr[8] = {0x44494250u, (uint32_t)p_GetCWBVHNodesSize(h), (uint32_t)p_GetCWBVHTrisSize(h), 8u, 3u, 1u,
                                 (uint32_t)passes, (uint32_t)sizeof(PTFrameParams)};
        fwrite(hdr, sizeof(hdr), 1, dump);
        fwrite(verts, sizeof(verts), 1, dump);
        fwrite(attrs, sizeof(attrs), 1, dump);
        fwrite(mats, sizeof(mats), 1, dump);
        fwrite(&light, sizeof(light), 1, dump);
        fwrite(nodes, hdr[1], 1, dump);
        fwrite(tris, hdr[2], 1, dump);
    }

    PTContext* ctx = NULL;
    PTGroup* group = NULL;
    int rc = nDevices ? p_PTCreateMulti(devices, nDevices, &group) : p_PTCreate(0, &ctx);
    if (rc) { fprintf(stderr, "PTCreate%s: %d %s\n", nDevices ? "Multi" : "", rc, p_PTGetLastError()); return 1; }
    PTSceneDesc sd;
    memset(&sd, 0, sizeof(sd));
    sd.structSize = (uint32_t)sizeof(sd);
    sd.bvhNodes = nodes; sd.bvhNodesBytes = (uint64_t)p_GetCWBVHNodesSize(h);
    sd.bvhTris = tris;   sd.bvhTrisBytes = (uint64_t)p_GetCWBVHTrisSize(h);
    sd.triAttrs = attrs; sd.triAttrsBytes = sizeof(attrs);
    sd.materials = mats; sd.materialCount = 3;
    sd.lights = &light;  sd.lightCount = 1;
    sd.features = PT_FEATURE_HAS_LIGHTS;
    rc = group ? p_PTGroupSetScene(group, &sd) : p_PTSetScene(ctx, &sd);
    if (rc) { fprintf(stderr, "PTSetScene: %d %s\n", rc, p_PTGetLastError()); return 1; }
    p_DestroyBVH(h);                                /* the C# host frees the BVH right after upload (BVHScene.cs:764-766) */

    PTFrameParams p;
    memset(&p, 0, sizeof(p));
    p.structSize = (uint32_t)sizeof(p);
    float t = tanf(40.0f * 0.5f * 3.14159265f / 180.0f), aspect = (float)W / (float)H, n = 0.1f, f = 100.0f;
    /* inverse of the GL projection, Matrix4x4 memory order (element (r,c) at c*4+r) */
    p.CamInvProj[0] = t * aspect; p.CamInvProj[5] = t; p.CamInvProj[11] = (n - f) / (2.0f * f * n);
    p.CamInvProj[14] = -1.0f;     p.CamInvProj[15] = (f + n) / (2.0f * f * n);
    p.CamToWorld[0] = 1; p.CamToWorld[5] = 1; p.CamToWorld[10] = -1; p.CamToWorld[15] = 1;   /* looks down -Z -> world +z */
    p.CamToWorld[13] = 1.0f; p.CamToWorld[14] = -3.4f;
    p.MaxRayBounces = 4; p.SamplesPerPass = spp; p.OutputWidth = (uint32_t)W; p.OutputHeight = (uint32_t)H;
    p.EnvironmentIntensity = 1.0f; p.EnvironmentColor[3] = 1.0f; p.UseRussianRoulette = 1;
    int currentSample = 0, maxSamples = 100000, flipped = 0;
    for (int k = 0; k < passes; k++) {              /* PathTracer.cs:224-272 */
        p.CurrentSample = (uint32_t)currentSample;
        p.RngSeedRoot = 0x12345678u + (uint32_t)k * 0x9E3779B9u;
        if (dump) fwrite(&p, sizeof(p), 1, dump);
        rc = group ? p_PTGroupRenderPass(group, &p) : p_PTRenderPass(ctx, &p);
        if (rc) { fprintf(stderr, "PTRenderPass: %d %s\n", rc, p_PTGetLastError()); return 1; }
        currentSample += spp > 1 ? spp : 1;
        flipped = 0;
        if (currentSample < maxSamples && k + 1 < passes) { if (group) p_PTGroupFlipFrames(group); else p_PTFlipFrames(ctx); flipped = 1; }
    }
    (void)flipped;
    float* frame = (float*)malloc((size_t)W * H * 4 * sizeof(float));
    rc = group ? p_PTGroupReadback(group, frame, (uint64_t)W * H * 4) : p_PTReadback(ctx, frame, (uint64_t)W * H * 4);
    if (rc) { fprintf(stderr, "PTReadback: %d %s\n", rc, p_PTGetLastError()); return 1; }
    PTStats st;
    if (group) p_PTGroupGetStats(group, &st); else p_PTGetStats(ctx, &st);
    double mean[3] = {0, 0, 0};
    for (long i = 0; i < (long)W * H; i++) for (int c = 0; c < 3; c++) mean[c] += frame[i * 4 + c];
    printf("pt_host: %dx%d, %d spp x %d passes, rays %llu, mean radiance %.5f %.5f %.5f\n", W, H, spp, passes,
           (unsigned long long)(st.closestHitRays + st.shadowRays), mean[0] / (W * H), mean[1] / (W * H), mean[2] / (W * H));
    if (out) { FILE* fo = fopen(out, "wb"); if (fo) { fwrite(frame, sizeof(float), (size_t)W * H * 4, fo); fclose(fo); } }
    free(frame);
    if (dump) fclose(dump);
    if (group) p_PTGroupDestroy(group); else p_PTDestroy(ctx);
    dlclose(lib);
    return 0;
}
