/*
 * ref_probe.c -- thin flat-C wrappers around the UNMODIFIED reference sources.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md).  This file contains no
 * reference code: it #includes three reference translation units where they
 * lie under /root/reference/src (the Makefile passes -I for that directory,
 * nothing is copied into this repository) so that their file-static functions
 * and private structs become reachable, and re-exports them with plain
 * pointer/size signatures for ctypes.  It is compiled only in the build
 * container (the GPU box has no /root/reference); outputs go to oracle/_ref/.
 *
 * Written in C89 because the reference is built with -std=c89 (Makefile:3)
 * and this translation unit is compiled with the reference's own flags.
 */
#include "SLAPredictor.c"
#include "SLACoder.c"
#include "SLAEncoder.c"
#include "SLADecoder.h"

#include "sla_flat.h"

/* ---- predictor-level wrappers ------------------------------------------ */

int ref_autocorr(const double* x, uint32_t n, double* r, uint32_t nlags)
{
  return (int)LPC_CalculateAutoCorrelation(x, n, r, nlags);
}

int ref_levinson(const double* r, uint32_t order, double* lpc, double* parcor)
{
  struct SLALPCCalculator* c = SLALPCCalculator_Create(order);
  int ret = (int)LPC_LevinsonDurbinRecursion(c, r, lpc, parcor, order);
  SLALPCCalculator_Destroy(c);
  return ret;
}

int ref_parcor(const double* x, uint32_t n, uint32_t order, double* parcor)
{
  struct SLALPCCalculator* c = SLALPCCalculator_Create(order);
  int ret = (int)SLALPCCalculator_CalculatePARCORCoefDouble(c, x, n, parcor, order);
  SLALPCCalculator_Destroy(c);
  return ret;
}

int ref_code_length(const double* x, uint32_t n, uint32_t bps,
    const double* parcor, uint32_t order, double* out)
{
  return (int)SLALPCCalculator_EstimateCodeLength(x, n, bps, parcor, order, out);
}

int ref_lattice_predict(const int32_t* x, uint32_t n, const int32_t* kint,
    uint32_t order, int32_t* res)
{
  struct SLALPCSynthesizer* s = SLALPCSynthesizer_Create(order);
  int ret = (int)SLALPCSynthesizer_PredictByParcorCoefInt32(s, x, n, kint, order, res);
  SLALPCSynthesizer_Destroy(s);
  return ret;
}

int ref_lattice_synth(const int32_t* res, uint32_t n, const int32_t* kint,
    uint32_t order, int32_t* out)
{
  struct SLALPCSynthesizer* s = SLALPCSynthesizer_Create(order);
  int ret = (int)SLALPCSynthesizer_SynthesizeByParcorCoefInt32(s, res, n, kint, order, out);
  SLALPCSynthesizer_Destroy(s);
  return ret;
}

int ref_preemph_i32(int32_t* data, uint32_t n)
{
  struct SLAEmphasisFilter* e = SLAEmphasisFilter_Create();
  int ret = (int)SLAEmphasisFilter_PreEmphasisInt32(e, data, n, SLA_PRE_EMPHASIS_COEFFICIENT_SHIFT);
  SLAEmphasisFilter_Destroy(e);
  return ret;
}

int ref_deemph_i32(int32_t* data, uint32_t n)
{
  struct SLAEmphasisFilter* e = SLAEmphasisFilter_Create();
  int ret = (int)SLAEmphasisFilter_DeEmphasisInt32(e, data, n, SLA_PRE_EMPHASIS_COEFFICIENT_SHIFT);
  SLAEmphasisFilter_Destroy(e);
  return ret;
}

void ref_preemph_f64(double* data, uint32_t n)
{
  SLAEmphasisFilter_PreEmphasisDouble(data, n, SLA_PRE_EMPHASIS_COEFFICIENT_SHIFT);
}

int ref_ltm_analyze(const int32_t* res, uint32_t n, uint32_t fft_size,
    uint32_t max_taps, uint32_t ntaps, uint32_t* pitch, double* coef, double* autocorr_out)
{
  struct SLALongTermCalculator* c = SLALongTermCalculator_Create(
      fft_size, SLALONGTERM_MAX_PERIOD, SLALONGTERM_NUM_PITCH_CANDIDATES, max_taps);
  int ret;
  *pitch = 0xFFFFFFFFU;
  ret = (int)SLALongTermCalculator_CalculateCoef(c, res, n, pitch, coef, ntaps);
  if (autocorr_out != NULL) {
    memcpy(autocorr_out, c->auto_corr, sizeof(double) * fft_size);
  }
  SLALongTermCalculator_Destroy(c);
  return ret;
}

int ref_ltm_predict(const int32_t* in, uint32_t n, uint32_t pitch,
    const int32_t* coef, uint32_t ntaps, int32_t* out)
{
  struct SLALongTermSynthesizer* s = SLALongTermSynthesizer_Create(ntaps, SLALONGTERM_MAX_PERIOD);
  int ret = (int)SLALongTermSynthesizer_PredictInt32(s, in, n, pitch, coef, ntaps, out);
  SLALongTermSynthesizer_Destroy(s);
  return ret;
}

int ref_ltm_synth(const int32_t* in, uint32_t n, uint32_t pitch,
    const int32_t* coef, uint32_t ntaps, int32_t* out)
{
  struct SLALongTermSynthesizer* s = SLALongTermSynthesizer_Create(ntaps, SLALONGTERM_MAX_PERIOD);
  int ret = (int)SLALongTermSynthesizer_SynthesizeInt32(s, in, n, pitch, coef, ntaps, out);
  SLALongTermSynthesizer_Destroy(s);
  return ret;
}

int ref_lms_predict(const int32_t* in, uint32_t n, uint32_t order, int32_t* out)
{
  struct SLALMSFilter* f = SLALMSFilter_Create(order);
  int ret = (int)SLALMSFilter_PredictInt32(f, order, in, n, out);
  SLALMSFilter_Destroy(f);
  return ret;
}

int ref_lms_synth(const int32_t* in, uint32_t n, uint32_t order, int32_t* out)
{
  struct SLALMSFilter* f = SLALMSFilter_Create(order);
  int ret = (int)SLALMSFilter_SynthesizeInt32(f, order, in, n, out);
  SLALMSFilter_Destroy(f);
  return ret;
}

/* data: planar [nch][n] doubles */
int ref_partition_search(const double* data, uint32_t nch, uint32_t n,
    uint32_t min_blk, uint32_t delta, uint32_t max_blk, uint32_t bps, uint32_t order,
    uint32_t* num_parts, uint32_t* parts)
{
  const double* ptr[SLA_MAX_CHANNELS];
  struct SLAOptimalBlockPartitionEstimator* oee;
  struct SLALPCCalculator* lpcc;
  uint32_t ch;
  int ret;
  for (ch = 0; ch < nch; ch++) { ptr[ch] = &data[(size_t)ch * n]; }
  oee  = SLAOptimalEncodeEstimator_Create((n > max_blk ? n : max_blk) > delta ? (n > max_blk ? n : max_blk) : delta, delta);
  lpcc = SLALPCCalculator_Create(order);
  ret = (int)SLAOptimalEncodeEstimator_SearchOptimalBlockPartitions(oee, lpcc,
      ptr, nch, n, min_blk, delta, max_blk, bps, order, num_parts, parts);
  SLALPCCalculator_Destroy(lpcc);
  SLAOptimalEncodeEstimator_Destroy(oee);
  return ret;
}

/* adjacency: row-major [nodes][nodes] */
int ref_dijkstra(const double* adjacency, uint32_t nodes, uint32_t start, uint32_t goal,
    double* min_cost, uint32_t* path)
{
  struct SLAOptimalBlockPartitionEstimator* oee
    = SLAOptimalEncodeEstimator_Create(1024 * (nodes > 1 ? nodes - 1 : 1), 1024);
  uint32_t i, j;
  int ret;
  for (i = 0; i < nodes; i++) {
    for (j = 0; j < nodes; j++) { oee->adjacency_matrix[i][j] = adjacency[i * nodes + j]; }
  }
  ret = (int)SLAOptimalEncodeEstimator_ApplyDijkstraMethod(oee, nodes, start, goal, min_cost);
  for (i = 0; i < nodes; i++) { path[i] = oee->path[i]; }
  SLAOptimalEncodeEstimator_Destroy(oee);
  return ret;
}

/* ---- utility wrappers --------------------------------------------------- */

uint32_t ref_crc16(const uint8_t* data, uint32_t n)
{
  return (uint32_t)SLAUtility_CalculateCRC16(data, n);
}

void ref_fft(double* data, uint32_t n, int32_t sign)
{
  SLAUtility_FFT(data, n, sign);
}

int ref_window(uint32_t type, double* w, uint32_t n)
{
  switch (type) {
    case SLA_WINDOWFUNCTIONTYPE_RECTANGULAR: SLAUtility_MakeRectangularWindow(w, n); break;
    case SLA_WINDOWFUNCTIONTYPE_SIN:         SLAUtility_MakeSinWindow(w, n);         break;
    case SLA_WINDOWFUNCTIONTYPE_HANN:        SLAUtility_MakeHannWindow(w, n);        break;
    case SLA_WINDOWFUNCTIONTYPE_BLACKMAN:    SLAUtility_MakeBlackmanWindow(w, n);    break;
    case SLA_WINDOWFUNCTIONTYPE_VORBIS:      SLAUtility_MakeVorbisWindow(w, n);      break;
    default: return -1;
  }
  return 0;
}

uint32_t ref_bitwidth(const int32_t* data, uint32_t n)
{
  return SLAUtility_GetDataBitWidth(data, n);
}

int ref_lesolve(const double* A, double* b, uint32_t dim, uint32_t iters)
{
  struct SLALESolver* s = SLALESolver_Create(dim);
  const double* rows[16];
  uint32_t i;
  int ret;
  for (i = 0; i < dim; i++) { rows[i] = &A[i * dim]; }
  ret = (int)SLALESolver_Solve(s, rows, b, dim, iters);
  SLALESolver_Destroy(s);
  return ret;
}

/* ---- coder wrappers ----------------------------------------------------- */

/* res: planar [nch][n]; rice_init: [nch] */
void ref_rice_init(const int32_t* res, uint32_t nch, uint32_t n, uint32_t* rice_init)
{
  const int32_t* ptr[SLA_MAX_CHANNELS];
  struct SLACoder* c = SLACoder_Create(nch, SLACODER_NUM_RECURSIVERICE_PARAMETER);
  uint32_t ch;
  for (ch = 0; ch < nch; ch++) { ptr[ch] = &res[(size_t)ch * n]; }
  SLACoder_CalculateInitialRecursiveRiceParameter(c, SLACODER_NUM_RECURSIVERICE_PARAMETER, ptr, nch, n);
  for (ch = 0; ch < nch; ch++) {
    rice_init[ch] = (uint32_t)SLACODER_PARAMETER_GET(c->init_rice_parameter[ch], 0);
  }
  SLACoder_Destroy(c);
}

/* Rice-code a residual array exactly as a block body is coded (init params
 * with `bps` bits per channel, flush, data array, flush). Returns bytes. */
uint32_t ref_code_residual(const int32_t* res, uint32_t nch, uint32_t n, uint32_t bps,
    uint8_t* out, uint32_t cap)
{
  const int32_t* ptr[SLA_MAX_CHANNELS];
  struct SLACoder* c = SLACoder_Create(nch, SLACODER_NUM_RECURSIVERICE_PARAMETER);
  struct SLABitStream strm;
  uint32_t ch;
  int32_t size;
  for (ch = 0; ch < nch; ch++) { ptr[ch] = &res[(size_t)ch * n]; }
  SLACoder_CalculateInitialRecursiveRiceParameter(c, SLACODER_NUM_RECURSIVERICE_PARAMETER, ptr, nch, n);
  SLABitWriter_Open(&strm, out, cap);
  for (ch = 0; ch < nch; ch++) {
    SLACoder_PutInitialRecursiveRiceParameter(c, &strm, SLACODER_NUM_RECURSIVERICE_PARAMETER, bps, ch);
  }
  SLABitStream_Flush(&strm);
  SLACoder_PutDataArray(c, &strm, SLACODER_NUM_RECURSIVERICE_PARAMETER, ptr, nch, n);
  SLABitStream_Flush(&strm);
  SLABitStream_Tell(&strm, &size);
  SLABitStream_Close(&strm);
  SLACoder_Destroy(c);
  return (uint32_t)size;
}

/* Inverse of ref_code_residual. */
void ref_decode_residual(const uint8_t* in, uint32_t size, uint32_t nch, uint32_t n, uint32_t bps,
    int32_t* res)
{
  int32_t* ptr[SLA_MAX_CHANNELS];
  struct SLACoder* c = SLACoder_Create(nch, SLACODER_NUM_RECURSIVERICE_PARAMETER);
  struct SLABitStream strm;
  uint32_t ch;
  for (ch = 0; ch < nch; ch++) { ptr[ch] = &res[(size_t)ch * n]; }
  SLABitReader_Open(&strm, (uint8_t*)in, size);
  for (ch = 0; ch < nch; ch++) {
    SLACoder_GetInitialRecursiveRiceParameter(c, &strm, SLACODER_NUM_RECURSIVERICE_PARAMETER, bps, ch);
  }
  SLABitStream_Flush(&strm);
  SLACoder_GetDataArray(c, &strm, SLACODER_NUM_RECURSIVERICE_PARAMETER, ptr, nch, n);
  SLABitStream_Close(&strm);
  SLACoder_Destroy(c);
}

/* ---- encoder / decoder wrappers ---------------------------------------- */

static struct SLAEncoder* probe_make_encoder(const sla_flat_params* p)
{
  struct SLAEncoderConfig cfg;
  struct SLAWaveFormat wf;
  struct SLAEncodeParameter ep;
  struct SLAEncoder* enc;

  cfg.max_num_channels         = p->cap_channels;
  cfg.max_num_block_samples    = p->cap_block_samples;
  cfg.max_parcor_order         = p->cap_parcor_order;
  cfg.max_longterm_order       = p->cap_longterm_order;
  cfg.max_lms_order_per_filter = p->cap_lms_order;
  cfg.verpose_flag             = 0;
  enc = SLAEncoder_Create(&cfg);
  if (enc == NULL) { return NULL; }

  wf.num_channels   = p->num_channels;
  wf.bit_per_sample = p->bits_per_sample;
  wf.sampling_rate  = p->sampling_rate;
  wf.offset_lshift  = 0;
  ep.parcor_order          = p->parcor_order;
  ep.longterm_order        = p->longterm_order;
  ep.lms_order_per_filter  = p->lms_order;
  ep.ch_process_method     = (SLAChannelProcessMethod)p->ch_process_method;
  ep.window_function_type  = (SLAWindowFunctionType)p->window_type;
  ep.max_num_block_samples = p->max_block_samples;
  if (SLAEncoder_SetWaveFormat(enc, &wf) != SLA_APIRESULT_OK
      || SLAEncoder_SetEncodeParameter(enc, &ep) != SLA_APIRESULT_OK) {
    SLAEncoder_Destroy(enc);
    return NULL;
  }
  return enc;
}

/* input: planar [C][n] left-justified int32 */
int ref_encode_whole(const sla_flat_params* p, const int32_t* input, uint32_t n,
    uint8_t* out, uint32_t cap, uint32_t* out_size)
{
  const int32_t* ptr[SLA_MAX_CHANNELS];
  struct SLAEncoder* enc = probe_make_encoder(p);
  uint32_t ch;
  int ret;
  if (enc == NULL) { return -1; }
  for (ch = 0; ch < p->num_channels; ch++) { ptr[ch] = &input[(size_t)ch * n]; }
  ret = (int)SLAEncoder_EncodeWhole(enc, ptr, n, out, cap, out_size);
  SLAEncoder_Destroy(enc);
  return ret;
}

/* Fixed-size EncodeBlock calls under a header (the 1024-sample
 * plumbing configuration is only reachable this way).  header_max_block is
 * what goes into SetEncodeParameter / the header; block_samples is the size
 * of every EncodeBlock call. */
int ref_encode_fixed_blocks(const sla_flat_params* p, const int32_t* input, uint32_t n,
    uint32_t block_samples, uint8_t* out, uint32_t cap, uint32_t* out_size)
{
  const int32_t* ptr[SLA_MAX_CHANNELS];
  struct SLAEncoder* enc = probe_make_encoder(p);
  struct SLAHeaderInfo header;
  uint32_t ch, pos, cur, bsize, nblocks, maxblk, maxbps, bps_blk;
  int ret;
  if (enc == NULL) { return -1; }
  header.wave_format  = enc->wave_format;
  header.encode_param = enc->encode_param;
  header.num_samples  = n;
  cur = SLA_HEADER_SIZE; nblocks = 0; maxblk = 0; maxbps = 0; ret = 0;
  for (pos = 0; pos < n; pos += block_samples) {
    uint32_t cnt = (n - pos < block_samples) ? (n - pos) : block_samples;
    for (ch = 0; ch < p->num_channels; ch++) { ptr[ch] = &input[(size_t)ch * n + pos]; }
    ret = (int)SLAEncoder_EncodeBlock(enc, ptr, cnt, &out[cur], cap - cur, &bsize);
    if (ret != 0) { break; }
    cur += bsize;
    if (bsize > maxblk) { maxblk = bsize; }
    bps_blk = (8 * bsize * enc->wave_format.sampling_rate) / cnt;
    if (bps_blk > maxbps) { maxbps = bps_blk; }
    nblocks++;
  }
  header.num_blocks = nblocks;
  header.max_block_size = maxblk;
  header.max_bit_per_second = maxbps;
  if (ret == 0) { ret = (int)SLAEncoder_EncodeHeader(&header, out, cap); }
  *out_size = cur;
  SLAEncoder_Destroy(enc);
  return ret;
}

/* Whole-file encode that snapshots the encoder's internal per-block state.
 * The driving loop below is this probe's own restatement of the loop in
 * SLAEncoder_EncodeWhole (src/SLAEncoder.c:846-901) calling the reference's
 * own static helpers; tests check its bytes against ref_encode_whole(). */
int ref_encode_trace(const sla_flat_params* p, const int32_t* input, uint32_t n,
    uint8_t* out, uint32_t cap, uint32_t* out_size, sla_flat_trace* tr)
{
  const int32_t* ptr[SLA_MAX_CHANNELS];
  struct SLAEncoder* enc = probe_make_encoder(p);
  struct SLAHeaderInfo header;
  uint32_t ch, pos, cur, nblocks, maxblk, maxbps, C, O;
  int ret = 0;

  if (enc == NULL) { return -1; }
  C = p->num_channels;
  O = p->parcor_order + 1;
  for (ch = 0; ch < C; ch++) { ptr[ch] = &input[(size_t)ch * n]; }

  header.wave_format    = enc->wave_format;
  header.encode_param   = enc->encode_param;
  header.num_samples    = n;
  header.max_block_size = SLA_MAX_BLOCK_SIZE_INVAILD;
  header.num_blocks = 0; header.max_bit_per_second = 0;
  if ((ret = (int)SLAEncoder_EncodeHeader(&header, out, cap)) != 0) { goto done; }
  header.wave_format.offset_lshift = enc->wave_format.offset_lshift
    = (uint8_t)SLAEncoder_CalculateLeftShiftOffset(enc, ptr, n);
  tr->offset_lshift = enc->wave_format.offset_lshift;

  cur = SLA_HEADER_SIZE; nblocks = 0; maxblk = 0; maxbps = 0; pos = 0;
  while (pos < n) {
    const int32_t* bptr[SLA_MAX_CHANNELS];
    uint32_t remain = n - pos, nparts, part;
    uint32_t win = SLAUTILITY_MIN(enc->encode_param.max_num_block_samples, remain);
    if (cur >= cap) { ret = SLA_APIRESULT_INSUFFICIENT_BUFFER_SIZE; goto done; }
    for (ch = 0; ch < C; ch++) { bptr[ch] = &input[(size_t)ch * n + pos]; }
    ret = (int)SLAEncoder_SearchOptimalBlockPartitions(enc, bptr, win,
        (uint32_t)SLAUTILITY_MIN(SLA_MIN_BLOCK_NUM_SAMPLES, remain),
        SLA_SEARCH_BLOCK_NUM_SAMPLES_DELTA, win, &nparts, enc->num_block_partition_samples);
    if (ret != 0) { goto done; }
    for (part = 0; part < nparts; part++) {
      uint32_t cnt = enc->num_block_partition_samples[part], bsize, bps_blk, ord;
      for (ch = 0; ch < C; ch++) { bptr[ch] = &input[(size_t)ch * n + pos]; }
      ret = (int)SLAEncoder_EncodeBlock(enc, bptr, cnt, &out[cur], cap - cur, &bsize);
      if (ret != 0) { goto done; }
      if (nblocks < tr->max_blocks) {
        uint32_t b = nblocks;
        tr->blk_start[b] = pos; tr->blk_nsmpl[b] = cnt;
        tr->blk_type[b] = (uint32_t)enc->block_data_type; tr->blk_bytes[b] = bsize;
        for (ch = 0; ch < C; ch++) {
          size_t bc = (size_t)b * C + ch;
          if (enc->block_data_type != SLA_BLOCK_DATA_TYPE_COMPRESSDATA) {
            tr->rshift[bc] = 0; tr->pitch[bc] = 0; tr->rice_init[bc] = 0;
            continue;
          }
          for (ord = 0; ord < O; ord++) {
            tr->parcor[bc * tr->order_stride + ord] = enc->parcor_coef[ch][ord];
            tr->code[bc * tr->order_stride + ord]   = (ord == 0) ? 0 : enc->parcor_coef_code[ch][ord];
            tr->kint[bc * tr->order_stride + ord]   = enc->parcor_coef_int32[ch][ord];
          }
          tr->rshift[bc] = enc->parcor_rshift[ch];
          tr->pitch[bc]  = enc->pitch_period[ch];
          for (ord = 0; ord < p->longterm_order; ord++) {
            tr->ltm_coef[bc * tr->ltm_stride + ord] = enc->longterm_coef_int32[ch][ord];
          }
          tr->rice_init[bc] = (uint32_t)SLACODER_PARAMETER_GET(enc->coder->init_rice_parameter[ch], 0);
          memcpy(&tr->res_final[(size_t)ch * tr->sample_stride + pos], enc->residual[ch], sizeof(int32_t) * cnt);
          /* lattice residual: replay pre-emphasis + lattice with the reference's own filters */
          {
            int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * cnt);
            memcpy(tmp, enc->input_int32[ch], sizeof(int32_t) * cnt);
            ref_preemph_i32(tmp, cnt);
            ref_lattice_predict(tmp, cnt, enc->parcor_coef_int32[ch], p->parcor_order,
                &tr->res_lattice[(size_t)ch * tr->sample_stride + pos]);
            free(tmp);
          }
        }
      }
      cur += bsize; pos += cnt;
      if (bsize > maxblk) { maxblk = bsize; }
      bps_blk = (8 * bsize * enc->wave_format.sampling_rate) / cnt;
      if (bps_blk > maxbps) { maxbps = bps_blk; }
      nblocks++;
    }
  }
  header.num_blocks = nblocks;
  header.max_block_size = maxblk;
  header.max_bit_per_second = maxbps;
  ret = (int)SLAEncoder_EncodeHeader(&header, out, cap);
  *out_size = cur;
  tr->num_blocks = nblocks;
done:
  SLAEncoder_Destroy(enc);
  return ret;
}

/* out: planar [C][nmax]; hdr_out: 12 uint32 = channels, bps, rate, lshift, parcor,
 * ltm, lms, chproc, num_samples, num_blocks, max_block_samples, max_block_size */
int ref_decode_whole(const sla_flat_params* p, const uint8_t* data, uint32_t size,
    int32_t* out, uint32_t nmax, uint32_t* nsamples, uint32_t* hdr_out)
{
  struct SLADecoderConfig cfg;
  struct SLADecoder* dec;
  struct SLAHeaderInfo hdr;
  int32_t* ptr[SLA_MAX_CHANNELS];
  uint32_t ch;
  int ret;

  ret = (int)SLADecoder_DecodeHeader(data, size, &hdr);
  if (ret != 0) { return ret; }
  if (hdr_out != NULL) {
    hdr_out[0] = hdr.wave_format.num_channels; hdr_out[1] = hdr.wave_format.bit_per_sample;
    hdr_out[2] = hdr.wave_format.sampling_rate; hdr_out[3] = hdr.wave_format.offset_lshift;
    hdr_out[4] = hdr.encode_param.parcor_order; hdr_out[5] = hdr.encode_param.longterm_order;
    hdr_out[6] = hdr.encode_param.lms_order_per_filter; hdr_out[7] = (uint32_t)hdr.encode_param.ch_process_method;
    hdr_out[8] = hdr.num_samples; hdr_out[9] = hdr.num_blocks;
    hdr_out[10] = hdr.encode_param.max_num_block_samples; hdr_out[11] = hdr.max_block_size;
  }
  cfg.max_num_channels         = p->cap_channels;
  cfg.max_num_block_samples    = p->cap_block_samples;
  cfg.max_parcor_order         = p->cap_parcor_order;
  cfg.max_longterm_order       = p->cap_longterm_order;
  cfg.max_lms_order_per_filter = p->cap_lms_order;
  cfg.enable_crc_check         = 1;
  cfg.verpose_flag             = 0;
  dec = SLADecoder_Create(&cfg);
  if (dec == NULL) { return -1; }
  for (ch = 0; ch < hdr.wave_format.num_channels; ch++) { ptr[ch] = &out[(size_t)ch * nmax]; }
  ret = (int)SLADecoder_DecodeWhole(dec, data, size, ptr, nmax, nsamples);
  SLADecoder_Destroy(dec);
  return ret;
}
