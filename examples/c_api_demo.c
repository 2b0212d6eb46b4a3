/*
 * Plain-C caller of libvitpose_hip.so: the drop-in boundary needs nothing but <stdint.h> types.
 *
 *   gcc -std=c99 -Iinclude examples/c_api_demo.c -o c_api_demo \
 *       -Leasy_vitpose_amd/_lib -lvitpose_hip -Wl,-rpath,$PWD/easy_vitpose_amd/_lib -lm
 *   ./c_api_demo            # needs an MI355X; without one every compute entry returns VP_ERR_HIP
 *
 * Builds a ViTPose-S / COCO-17 handle with pseudo-random weights (schema of the reference's state dict),
 * runs 4 uint8 crops through vp_infer and prints the first keypoints.  tests/test_host_logic.py compiles and links
 * this file on the CPU-only build box (no run) so that the header stays valid C.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vitpose_hip.h"

enum { D = 384, L = 12, H = 12, K = 17, N = 4 };

static uint32_t rng_state = 12345u;
static float rnd(float scale) {   /* xorshift, uniform in [-scale, scale) */
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 17; rng_state ^= rng_state << 5;
    return ((float)(rng_state >> 8) / 8388608.0f - 1.0f) * scale;
}

typedef struct { vp_tensor_desc* v; int n, cap; } tensor_list;

static void add(tensor_list* tl, const char* name, int64_t numel, float scale, float offset) {
    float* p = (float*)malloc((size_t)numel * sizeof(float));
    for (int64_t i = 0; i < numel; ++i) p[i] = offset + rnd(scale);
    if (tl->n == tl->cap) { tl->cap = tl->cap ? 2 * tl->cap : 64; tl->v = (vp_tensor_desc*)realloc(tl->v, (size_t)tl->cap * sizeof(*tl->v)); }
    char* nm = (char*)malloc(strlen(name) + 1);
    strcpy(nm, name);
    tl->v[tl->n].name = nm; tl->v[tl->n].data = p; tl->v[tl->n].numel = numel; tl->n++;
}

int main(void) {
    vp_config cfg = {D, L, H, K, VP_DTYPE_F16, 0, N};
    vp_handle h = NULL;
    int rc = vp_create(&h, &cfg);
    if (rc != VP_OK) { fprintf(stderr, "vp_create: status %d: %s\n", rc, vp_last_error(NULL)); return 2; }

    tensor_list tl = {0};
    char nm[128];
    add(&tl, "backbone.pos_embed", 193 * D, 0.03f, 0.f);
    add(&tl, "backbone.patch_embed.proj.weight", (int64_t)D * 3 * 16 * 16, 0.03f, 0.f);
    add(&tl, "backbone.patch_embed.proj.bias", D, 0.03f, 0.f);
    for (int i = 0; i < L; ++i) {
        const struct { const char* s; int64_t n; float sc, off; } t[] = {
            {"norm1.weight", D, 0.1f, 1.f}, {"norm1.bias", D, 0.05f, 0.f},
            {"attn.qkv.weight", 3LL * D * D, 0.06f, 0.f}, {"attn.qkv.bias", 3 * D, 0.03f, 0.f},
            {"attn.proj.weight", (int64_t)D * D, 0.03f, 0.f}, {"attn.proj.bias", D, 0.03f, 0.f},
            {"norm2.weight", D, 0.1f, 1.f}, {"norm2.bias", D, 0.05f, 0.f},
            {"mlp.fc1.weight", 4LL * D * D, 0.03f, 0.f}, {"mlp.fc1.bias", 4 * D, 0.03f, 0.f},
            {"mlp.fc2.weight", 4LL * D * D, 0.03f, 0.f}, {"mlp.fc2.bias", D, 0.03f, 0.f}};
        for (unsigned j = 0; j < sizeof(t) / sizeof(t[0]); ++j) {
            snprintf(nm, sizeof nm, "backbone.blocks.%d.%s", i, t[j].s);
            add(&tl, nm, t[j].n, t[j].sc, t[j].off);
        }
    }
    add(&tl, "backbone.last_norm.weight", D, 0.1f, 1.f);
    add(&tl, "backbone.last_norm.bias", D, 0.05f, 0.f);
    add(&tl, "keypoint_head.deconv_layers.0.weight", (int64_t)D * 256 * 16, 0.05f, 0.f);
    add(&tl, "keypoint_head.deconv_layers.3.weight", 256LL * 256 * 16, 0.06f, 0.f);
    for (int idx = 1; idx <= 4; idx += 3) {
        snprintf(nm, sizeof nm, "keypoint_head.deconv_layers.%d.weight", idx);       add(&tl, nm, 256, 0.1f, 1.f);
        snprintf(nm, sizeof nm, "keypoint_head.deconv_layers.%d.bias", idx);         add(&tl, nm, 256, 0.1f, 0.f);
        snprintf(nm, sizeof nm, "keypoint_head.deconv_layers.%d.running_mean", idx); add(&tl, nm, 256, 0.1f, 0.f);
        snprintf(nm, sizeof nm, "keypoint_head.deconv_layers.%d.running_var", idx);  add(&tl, nm, 256, 0.4f, 1.f);
    }
    add(&tl, "keypoint_head.final_layer.weight", K * 256, 0.03f, 0.f);
    add(&tl, "keypoint_head.final_layer.bias", K, 0.03f, 0.f);
    rc = vp_load_weights(h, tl.v, tl.n);
    if (rc != VP_OK) { fprintf(stderr, "vp_load_weights: status %d: %s\n", rc, vp_last_error(h)); vp_destroy(h); return 3; }

    uint8_t* crops = (uint8_t*)malloc((size_t)N * 256 * 192 * 3);
    for (size_t i = 0; i < (size_t)N * 256 * 192 * 3; ++i) crops[i] = (uint8_t)(rnd(128.f) + 128.f);
    int32_t org_wh[N][2];
    for (int i = 0; i < N; ++i) { org_wh[i][0] = 192; org_wh[i][1] = 256; }
    float* out = (float*)malloc((size_t)N * K * 3 * sizeof(float));
    rc = vp_infer(h, crops, VP_INPUT_U8_NHWC, N, &org_wh[0][0], out);
    if (rc != VP_OK) { fprintf(stderr, "vp_infer: status %d: %s\n", rc, vp_last_error(h)); vp_destroy(h); return 4; }
    for (int k = 0; k < 3; ++k) printf("crop 0 joint %d: y %.2f x %.2f conf %.4f\n", k, out[k * 3], out[k * 3 + 1], out[k * 3 + 2]);
    vp_destroy(h);
    return 0;
}
