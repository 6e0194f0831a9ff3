/* CPU ORACLE — TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * Plain-C restatement of the reference's only native component, the greedy per-class NMS in
 * /root/reference/cython_nms.pyx:24-74 (itself derived from Fast R-CNN's cpu_nms).  The .pyx does
 * not cythonize in this image (np.int_t / np.int were removed from numpy 2.x), so
 * there is no runnable reference for this function: PARITY UNPINNED by the reference; the
 * known-answer cases in tests/test_oracle_greedy_nms.py are hand-derived.
 *
 * Semantics restated line by line:
 *   :32      areas = (x2 - x1 + 1) * (y2 - y1 + 1)                      (float32, "+1" pixel convention)
 *   :33      order = scores.argsort()[::-1]                              (numpy quicksort, reversed)
 *   :49-72   for each box in score order, skip if suppressed, else suppress every later box with
 *            inter / (area_i + area_j - inter) >= thresh                 (float32 arithmetic, `>=`)
 *   :74      return np.where(suppressed == 0)[0]                         (kept indices, ASCENDING index order)
 *
 * Tie order: numpy's default argsort is not stable, so equal scores have no defined order in the
 * reference; this restatement (and the HIP kernel) break ties by HIGHER original index first, which
 * is what `argsort()[::-1]` gives for a stable ascending sort.
 */
#include <stdint.h>
#include <stdlib.h>

typedef struct { float s; int64_t i; } key_t_;

static int cmp_desc(const void* a, const void* b) {
    const key_t_* x = (const key_t_*)a; const key_t_* y = (const key_t_*)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i > y->i) ? -1 : (x->i < y->i);   /* reversed stable-ascending => higher index first */
}

/* dets: [n][5] float32 (x1,y1,x2,y2,score). keep_out: int64[n]. returns number kept. */
int oracle_greedy_nms(const float* dets, int n, float thresh, int64_t* keep_out) {
    if (n <= 0) return 0;
    key_t_* order = (key_t_*)malloc(sizeof(key_t_) * (size_t)n);
    float* area = (float*)malloc(sizeof(float) * (size_t)n);
    unsigned char* dead = (unsigned char*)calloc((size_t)n, 1);
    for (int i = 0; i < n; ++i) {
        const float* d = dets + 5 * (size_t)i;
        volatile float w = d[2] - d[0] + 1.0f, h = d[3] - d[1] + 1.0f;   /* volatile: no fma / excess precision */
        area[i] = w * h;
        order[i].s = d[4]; order[i].i = i;
    }
    qsort(order, (size_t)n, sizeof(key_t_), cmp_desc);
    for (int a = 0; a < n; ++a) {
        int i = (int)order[a].i;
        if (dead[i]) continue;
        const float* di = dets + 5 * (size_t)i;
        for (int b = a + 1; b < n; ++b) {
            int j = (int)order[b].i;
            if (dead[j]) continue;
            const float* dj = dets + 5 * (size_t)j;
            float xx1 = di[0] >= dj[0] ? di[0] : dj[0];
            float yy1 = di[1] >= dj[1] ? di[1] : dj[1];
            float xx2 = di[2] <= dj[2] ? di[2] : dj[2];
            float yy2 = di[3] <= dj[3] ? di[3] : dj[3];
            volatile float w = xx2 - xx1 + 1.0f, h = yy2 - yy1 + 1.0f;
            float ww = w >= 0.0f ? w : 0.0f, hh = h >= 0.0f ? h : 0.0f;
            volatile float inter = ww * hh;
            volatile float uni = area[i] + area[j];
            uni = uni - inter;
            float ovr = inter / uni;
            if (ovr >= thresh) dead[j] = 1;
        }
    }
    int k = 0;
    for (int i = 0; i < n; ++i) if (!dead[i]) keep_out[k++] = i;
    free(order); free(area); free(dead);
    return k;
}
