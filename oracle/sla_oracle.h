/*
 * sla_oracle.h -- CPU restatement of the SLA encode path (and its inverse).
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity checker for the HIP path in
 * sla_amd/: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it; the product never links or calls it.  Pinned against the
 * unmodified reference (oracle/_ref, built in the build container) by
 * tests/test_oracle_vs_ref.py and against committed vectors in tests/golden/.
 */
#ifndef SLA_ORACLE_H_INCLUDED
#define SLA_ORACLE_H_INCLUDED

#include <stdint.h>
#include "sla_flat.h"

#ifdef __cplusplus
extern "C" {
#endif

/* result codes follow SLAApiResult (reference src/include/public/SLA.h:26-43) */
enum {
  SLAO_OK = 0, SLAO_NG = 1, SLAO_INVALID_ARGUMENT = 2, SLAO_EXCEED_HANDLE_CAPACITY = 3,
  SLAO_INSUFFICIENT_BUFFER_SIZE = 4, SLAO_INVALID_CHPROCESSMETHOD = 5,
  SLAO_FAILED_TO_CALCULATE_COEF = 6, SLAO_FAILED_TO_PREDICT = 7, SLAO_FAILED_TO_SYNTHESIZE = 8,
  SLAO_INSUFFICIENT_DATA_SIZE = 9, SLAO_INVALID_HEADER_FORMAT = 10, SLAO_DETECT_DATA_CORRUPTION = 11,
  SLAO_FAILED_TO_FIND_SYNC_CODE = 12, SLAO_INVALID_WINDOWFUNCTION_TYPE = 13,
  SLAO_NO_DATA_FRAGMENTS = 14, SLAO_PARAMETER_NOT_SET = 15
};

/* unit level (same flat signatures as the ref_* probe) */
int      slao_autocorr(const double* x, uint32_t n, double* r, uint32_t nlags);
int      slao_levinson(const double* r, uint32_t order, double* lpc, double* parcor);
int      slao_parcor(const double* x, uint32_t n, uint32_t order, double* parcor);
int      slao_code_length(const double* x, uint32_t n, uint32_t bps, const double* parcor, uint32_t order, double* out);
int      slao_lattice_predict(const int32_t* x, uint32_t n, const int32_t* kint, uint32_t order, int32_t* res);
int      slao_lattice_synth(const int32_t* res, uint32_t n, const int32_t* kint, uint32_t order, int32_t* out);
int      slao_preemph_i32(int32_t* data, uint32_t n);
int      slao_deemph_i32(int32_t* data, uint32_t n);
void     slao_preemph_f64(double* data, uint32_t n);
int      slao_ltm_analyze(const int32_t* res, uint32_t n, uint32_t fft_size, uint32_t max_taps, uint32_t ntaps,
                          uint32_t* pitch, double* coef, double* autocorr_out);
int      slao_ltm_predict(const int32_t* in, uint32_t n, uint32_t pitch, const int32_t* coef, uint32_t ntaps, int32_t* out);
int      slao_ltm_synth(const int32_t* in, uint32_t n, uint32_t pitch, const int32_t* coef, uint32_t ntaps, int32_t* out);
int      slao_lms_predict(const int32_t* in, uint32_t n, uint32_t order, int32_t* out);
int      slao_lms_synth(const int32_t* in, uint32_t n, uint32_t order, int32_t* out);
int      slao_partition_search(const double* data, uint32_t nch, uint32_t n, uint32_t min_blk, uint32_t delta,
                               uint32_t max_blk, uint32_t bps, uint32_t order, uint32_t* num_parts, uint32_t* parts);
int      slao_dijkstra(const double* adjacency, uint32_t nodes, uint32_t start, uint32_t goal, double* min_cost, uint32_t* path);
uint32_t slao_crc16(const uint8_t* data, uint32_t n);
void     slao_fft(double* data, uint32_t n, int32_t sign);
int      slao_window(uint32_t type, double* w, uint32_t n);
uint32_t slao_bitwidth(const int32_t* data, uint32_t n);
int      slao_lesolve(const double* A, double* b, uint32_t dim, uint32_t iters);
void     slao_rice_init(const int32_t* res, uint32_t nch, uint32_t n, uint32_t* rice_init);
uint32_t slao_code_residual(const int32_t* res, uint32_t nch, uint32_t n, uint32_t bps, uint8_t* out, uint32_t cap);
void     slao_decode_residual(const uint8_t* in, uint32_t size, uint32_t nch, uint32_t n, uint32_t bps, int32_t* res);

/* codec level */
int slao_encode_whole(const sla_flat_params* p, const int32_t* input, uint32_t n,
                      uint8_t* out, uint32_t cap, uint32_t* out_size);
/* a range of a longer file: offset_lshift is the whole file's (the multi-GPU sharding tests assemble ranges) */
int slao_encode_range(const sla_flat_params* p, const int32_t* input, uint32_t n, uint32_t file_lshift,
                      uint8_t* out, uint32_t cap, uint32_t* out_size);
int slao_encode_fixed_blocks(const sla_flat_params* p, const int32_t* input, uint32_t n, uint32_t block_samples,
                             uint8_t* out, uint32_t cap, uint32_t* out_size);
int slao_encode_trace(const sla_flat_params* p, const int32_t* input, uint32_t n,
                      uint8_t* out, uint32_t cap, uint32_t* out_size, sla_flat_trace* tr);
int slao_decode_whole(const sla_flat_params* p, const uint8_t* data, uint32_t size,
                      int32_t* out, uint32_t nmax, uint32_t* nsamples, uint32_t* hdr_out);

#ifdef __cplusplus
}
#endif

#endif
