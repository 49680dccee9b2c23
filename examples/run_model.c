/* run_model.c -- the two C-ABIs from plain C (C99): build a model from the reference's own
 * command-line tokens (include/r3d_host.h), run it on the GPUs (include/r3d.h), write the
 * reference's output files.
 *
 *   gcc -std=c99 -Iinclude examples/run_model.c -Lradiative3d_amd/lib -lr3d_host -lr3d_hip \
 *       -Wl,-rpath,$PWD/radiative3d_amd/lib -o run_model
 *   ./run_model 1 out --grid-compiled=40 --toa-degree=5 --num-phonons=1M --seis-p2p=...
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "r3d_host.h"

int main(int argc, char** argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: %s N_GPUS OUTDIR <Radiative3D options...>\n", argv[0]);
    return 2;
  }
  const int n_gpus = atoi(argv[1]);
  const char* outdir = argv[2];
  r3dh_model* m = r3dh_model_from_args(argc - 3, (const char* const*)(argv + 3));
  if (!m) {
    fprintf(stderr, "model: %s\n", r3dh_last_error());
    return 1;
  }
  const r3d_model_desc* d = r3dh_model_desc(m);
  const size_t bins = (size_t)d->n_seismometers * d->params.n_bins;
  r3d_result res;
  memset(&res, 0, sizeof res);
  res.energy = (double*)calloc(bins ? bins * R3D_N_ENERGY : 1, sizeof(double));
  res.counts = (uint64_t*)calloc(bins ? bins * R3D_N_COUNT : 1, sizeof(uint64_t));
  /* a node: one engine per device, kept for as many runs as the host wants; the shards' blocks are summed on the
   * devices (RCCL) and read once (r3d_run_model(d, n, 0, seed, n_gpus, &res) is this in one call) */
  int devices[64];
  if (n_gpus < 1 || n_gpus > 64) {
    fprintf(stderr, "N_GPUS must be 1 .. 64\n");
    return 2;
  }
  for (int g = 0; g < n_gpus; g++) devices[g] = g;
  r3d_node* node = r3d_node_create(d, devices, n_gpus);
  if (!node) {
    fprintf(stderr, "node: %s\n", r3d_last_error());
    return 1;
  }
  if (r3d_node_run(node, r3dh_num_phonons(m), 0, r3dh_seed(m), &res)) {
    fprintf(stderr, "run: %s\n", r3d_last_error());
    return 1;
  }
  printf("shards: %d (summed by %s)\n", r3d_node_size(node), r3d_node_reduction(node));
  r3d_node_destroy(node);
  char trace[1024];
  snprintf(trace, sizeof trace, "%s/seis_traces_asc.dat", outdir);
  const char* summary = r3dh_write_outputs(m, &res, outdir, trace, NULL);
  if (!summary) {
    fprintf(stderr, "output: %s\n", r3dh_last_error());
    return 1;
  }
  fputs(summary, stdout);
  printf("histories: lost %llu timeout %llu invalid %llu\n", (unsigned long long)res.n_lost,
         (unsigned long long)res.n_timeout, (unsigned long long)res.n_invalid);
  free(res.energy), free(res.counts);
  r3dh_model_free(m);
  return 0;
}
