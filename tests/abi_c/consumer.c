/* A plain-C consumer of include/dg16.h (what a cgo / Rust `extern "C"` binding sees): parses a snarkjs-layout zkey,
 * decodes a proof.bin-style proof and verifies it against the key's own verification key -- host-side entry points
 * only, so it runs without a GPU; and shows that creating a context without one fails with a status, not a crash.
 * usage: consumer <file.zkey> <proof.bin> <public.bin (n x 32 B canonical LE)>  -> prints "accepted=0|1" */
#include <stdio.h>
#include <stdlib.h>

#include "dg16.h"

static void *slurp(const char *path, size_t *n) {
  FILE *f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END);
  *n = (size_t)ftell(f);
  fseek(f, 0, SEEK_SET);
  void *p = malloc(*n ? *n : 1);
  if (fread(p, 1, *n, f) != *n) { perror("read"); exit(2); }
  fclose(f);
  return p;
}

int main(int argc, char **argv) {
  if (argc != 4) return 2;
  size_t zn, pn, un;
  void *zraw = slurp(argv[1], &zn), *praw = slurp(argv[2], &pn), *pub = slurp(argv[3], &un);
  dg16_zkey *z = NULL;
  if (dg16_zkey_parse(zraw, zn, &z) != DG16_OK) { printf("zkey: %s\n", dg16_io_error()); return 1; }
  dg16_zkey_header h;
  dg16_zkey_header_get(z, &h);
  const void *pt[7];
  size_t cnt[7];
  int which[7] = {DG16_ZKEY_ALPHA_G1, DG16_ZKEY_BETA_G2, DG16_ZKEY_GAMMA_G2, DG16_ZKEY_DELTA_G2, DG16_ZKEY_IC,
                  DG16_ZKEY_A, DG16_ZKEY_H};
  for (int i = 0; i < 7; i++)
    if (dg16_zkey_points(z, which[i], &pt[i], &cnt[i]) != DG16_OK) return 1;
  printf("n_vars=%u n_public=%u domain=%u constraints=%u ic=%zu a=%zu h=%zu\n", h.n_vars, h.n_public, h.domain_size,
         h.num_constraints, cnt[4], cnt[5], cnt[6]);
  if (pn != 128) return 1;
  unsigned long long proof[32];
  if (dg16_proof_decompress(DG16_BN254, praw, 1, proof) != DG16_OK) { printf("proof: %s\n", dg16_serialize_error()); return 1; }
  int ok = -1;
  int rc = dg16_groth16_verify(DG16_BN254, pt[0], pt[1], pt[2], pt[3], pt[4], cnt[4], pub, un / 32, proof, 0, &ok);
  if (rc != DG16_OK) { printf("verify: status %d %s\n", rc, dg16_verify_error()); return 1; }
  printf("accepted=%d\n", ok);
  dg16_zkey_free(z);
  /* the container walk of an arkworks key file is host code too: a truncated file is a status, not a crash */
  {
    unsigned char tiny[40] = {0};
    dg16_arkkey_layout_t lay;
    rc = dg16_arkkey_layout(tiny, sizeof tiny, 1, &lay);
    printf("arkkey_layout=%d (%s)\n", rc, rc ? dg16_codec_error() : "ok");
  }
  /* compute entry points need a GPU: without one the library reports it */
  dg16_ctx *ctx = NULL;
  rc = dg16_ctx_create(0, &ctx);
  printf("ctx_create=%d\n", rc);
  if (ctx) dg16_ctx_destroy(ctx);
  free(zraw); free(praw); free(pub);
  return 0;
}
