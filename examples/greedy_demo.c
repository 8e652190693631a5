/* Minimal C caller of the engine's C ABI (include/hctr_hip.h): load a checkpoint exported as raw tensors, run the
 * fused greedy path on one uint8 line image and print the label ids. Plain C99, no Python, no torch:
 *
 *   gcc -std=c99 -I include examples/greedy_demo.c -L handwritten-chinese-ocr-samples_amd -lhctr_hip \
 *       -Wl,-rpath,$PWD/handwritten-chinese-ocr-samples_amd -o greedy_demo
 *   ./greedy_demo weights.bin 7358 line.u8 2000
 *
 * weights.bin: for each of the 254 state-dict entries (models/handwritten_ctr_model.py; test.py:152-153):
 *   u32 key length, key bytes, u32 dtype (1 = float32, 2 = int64), u32 ndim, i64 shape[ndim], raw data.
 * tools/export_weights.py writes this file from a .pth.tar checkpoint or the synthetic generator.
 * line.u8: 128 x W grey bytes (already resized; use hctr_resize_lines for arbitrary images). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hctr_hip.h"

static int fail(hctr_ctx* ctx, const char* what, int rc) {
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, hctr_last_error(ctx));
    return 1;
}

int main(int argc, char** argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: %s weights.bin num_classes line.u8 width\n", argv[0]);
        return 2;
    }
    const int num_classes = atoi(argv[2]), W = atoi(argv[4]);
    hctr_ctx* ctx = NULL;
    int rc = hctr_create(&ctx, 0, num_classes);
    if (rc) return fail(NULL, "hctr_create", rc);

    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    for (;;) {
        unsigned klen, dtype, ndim;
        char key[256];
        int64_t shape[8], n = 1;
        if (fread(&klen, 4, 1, f) != 1) break;                      /* end of file */
        if (klen >= sizeof key || fread(key, 1, klen, f) != klen) return 1;
        key[klen] = 0;
        if (fread(&dtype, 4, 1, f) != 1 || fread(&ndim, 4, 1, f) != 1 || ndim > 8) return 1;
        if (ndim && fread(shape, 8, ndim, f) != ndim) return 1;
        for (unsigned i = 0; i < ndim; ++i) n *= shape[i];
        const size_t bytes = (size_t)n * (dtype == 2 ? 8 : 4);
        void* data = malloc(bytes ? bytes : 1);
        if (!data || fread(data, 1, bytes, f) != bytes) return 1;
        rc = hctr_load_tensor(ctx, key, data, shape, (int)ndim, (int)dtype);   /* the library copies the data */
        free(data);
        if (rc) return fail(ctx, key, rc);
    }
    fclose(f);
    if ((rc = hctr_finalize_weights(ctx)) != 0) return fail(ctx, "hctr_finalize_weights", rc);

    unsigned char* img = (unsigned char*)malloc((size_t)128 * W);
    f = fopen(argv[3], "rb");
    if (!f || !img || fread(img, 1, (size_t)128 * W, f) != (size_t)128 * W) { perror(argv[3]); return 1; }
    fclose(f);
    int32_t* labels = (int32_t*)malloc(sizeof(int32_t) * (size_t)W);
    int32_t length = 0, width = W;
    rc = hctr_greedy(ctx, img, 0 /* uint8 */, 0 /* host memory */, &width, 1, W, labels, &length);
    if (rc) return fail(ctx, "hctr_greedy", rc);
    printf("%d labels:", (int)length);
    for (int i = 0; i < length; ++i) printf(" %d", (int)labels[i]);
    printf("\n");
    free(labels);
    free(img);
    hctr_destroy(ctx);
    return 0;
}
