/* CPU restatement of the generate job's exhaustive inner-product search.  TEST INFRASTRUCTURE.
 *
 * Reference call sites: SimANS/co_training/co_training_generate.py:359-384 (faiss.IndexFlatIP sharded over the GPUs
 * by index_cpu_to_all_gpus, add(passage_embedding.astype(float32))) and :415-421 (index.search(q, 200 | 1000)).
 * The algorithm lives in a third-party dependency that is absent from /root/reference and from this image
 * (faiss-gpu, version not pinned by the reference: "conda install faiss-gpu", SimANS/README.md:102-104), so its
 * published behaviour is restated: score[q][c] = sum_h q[h] * c[h] in float32, the k highest scores per query in
 * descending order.  Two things FAISS leaves unspecified are pinned HERE so that the index output can be compared
 * bit for bit: (1) the summation order -- one float32 fused-multiply-add chain in ascending h; (2) ties -- equal
 * scores are ordered by ascending passage id.  Parity with FAISS itself is therefore "unpinned" for exact ties and
 * last-ulp score differences.
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off oracle/topk_ref.c -o oracle/_cbuild/libsimx_oracle.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void simx_oracle_scores(int nq, int nc, int H, const float* q, const float* c, float* out /* [nq,nc] */) {
  for (int i = 0; i < nq; ++i)
    for (int j = 0; j < nc; ++j) {
      float s = 0.0f;
      for (int h = 0; h < H; ++h) s = fmaf(q[(size_t)i * H + h], c[(size_t)j * H + h], s);
      out[(size_t)i * nc + j] = s;
    }
}

typedef struct { float s; int64_t id; } pair_t;

static int cmp_desc(const void* a, const void* b) {
  const pair_t* x = (const pair_t*)a; const pair_t* y = (const pair_t*)b;
  if (x->s > y->s) return -1;
  if (x->s < y->s) return 1;
  return (x->id > y->id) - (x->id < y->id);
}

/* top-k of explicit (score, id) candidates per query; rows shorter than k are padded with (-inf, -1). */
void simx_oracle_topk(int nq, int m, const float* scores /* [nq,m] */, const int64_t* ids /* [nq,m] or NULL: id = base + j */,
                      int64_t id_base, int k, float* out_s /* [nq,k] */, int64_t* out_i /* [nq,k] */) {
  pair_t* buf = (pair_t*)malloc(sizeof(pair_t) * (size_t)(m > 0 ? m : 1));
  for (int i = 0; i < nq; ++i) {
    int n = 0;
    for (int j = 0; j < m; ++j) {
      const int64_t id = ids ? ids[(size_t)i * m + j] : id_base + j;
      if (id < 0) continue;                                        /* padding entries of a previous round */
      buf[n].s = scores[(size_t)i * m + j]; buf[n].id = id; ++n;
    }
    qsort(buf, (size_t)n, sizeof(pair_t), cmp_desc);
    for (int j = 0; j < k; ++j) {
      out_s[(size_t)i * k + j] = j < n ? buf[j].s : -INFINITY;
      out_i[(size_t)i * k + j] = j < n ? buf[j].id : -1;
    }
  }
  free(buf);
}
