/*
 * rag_hip.h - C ABI of librag_hip.so, the MI355X (gfx950) retrieval kernels.
 *
 * The reference (Luyu-H/Domain-RAG-System) is pure Python and has no FFI of its own;
 * the seam these entry points sit behind is its Python class API.  Each entry point
 * names the reference call it replaces.  The Python host classes in
 * domain-rag-system_amd/ bind this file through ctypes.
 *
 * Conventions
 *  - every function returns 0 on success, a negative RAG_E* code otherwise;
 *    rag_last_error() returns a thread-local message for the last failure.
 *  - handles are opaque pointers; they are immutable after create and may be searched
 *    from several host threads as long as each call uses its own stream and workspace
 *    (calls on one handle share that handle's workspace: serialise them or use one
 *    handle per thread).
 *  - "_dev" entry points take DEVICE pointers, enqueue on `stream` (a hipStream_t passed
 *    as void*, NULL = the default stream) and return without synchronising.  They may
 *    grow the handle's cached workspace (hipMalloc) when a call needs more than any
 *    earlier one; steady-state calls allocate nothing.
 *  - entry points without the suffix take HOST pointers, copy in/out and synchronise.
 *  - rows are reported as GLOBAL row ids: local row + the `row_base` given at create
 *    (row_base is the first row of this GPU's shard; 0 when unsharded).
 *  - result lists are ordered best first; ties by row id DESCENDING (the order that
 *    numpy's stable argsort followed by [::-1] gives; the reference's own order among
 *    equal scores is implementation defined).  Unused slots hold row -1.
 */
#ifndef RAG_HIP_H
#define RAG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAG_OK            0
#define RAG_EINVAL       -1   /* bad argument                                  */
#define RAG_EHIP         -2   /* a HIP runtime call failed                     */
#define RAG_ENOMEM       -3   /* allocation failed                             */
#define RAG_ENODEV       -4   /* no usable gfx950 device                       */
#define RAG_ELIMIT       -5   /* argument beyond a documented limit            */

#define RAG_METRIC_COSINE 0   /* rows are expected L2-normalised; score = dot  */
#define RAG_METRIC_DOT    1
#define RAG_METRIC_EUCLID 2   /* score = Euclidean distance, smaller is better */

#define RAG_MEM_HOST_COPY   0 /* pointer is host memory, the library copies it */
#define RAG_MEM_DEVICE_COPY 1 /* pointer is device memory, the library copies  */
#define RAG_MEM_DEVICE_BORROW 2 /* device memory, caller keeps it alive        */

#define RAG_MAX_K        1024 /* largest k of any top-k entry point            */
#define RAG_MAX_FUSE_LIST 512 /* entries per input list of the fusion kernels  */

typedef struct rag_dense_s* rag_dense_t;
typedef struct rag_bm25_s*  rag_bm25_t;
typedef struct rag_tokenized_s* rag_tokenized_t;

/* ---- library ----------------------------------------------------------------- */

/* Select `device` for the calling thread and check it is a gfx950 part. */
int rag_init(int device);
const char* rag_last_error(void);
int rag_version(void);

/* Device-side consistency checks (no counterpart in the reference).  No kernel indexes memory
 * with an unchecked value that it read from device data (row ids, term ids, offsets, list
 * lengths): a value outside its array sets one of these bits in a per-device word and the
 * kernel carries on with a safe substitute (the entry is skipped or treated as empty), so bad
 * input or an internal error yields flagged results instead of a GPU memory fault.
 * rag_fault_flags synchronises the current device, returns the bits collected since the last
 * clear (0 = every check held) and clears them when `clear` is non-zero.                 */
#define RAG_FAULT_DENSE_LIST    1u  /* dense scan: a candidate list outgrew its capacity     */
#define RAG_FAULT_MERGE_SHORT   2u  /* list merge: selection did not yield exactly k entries  */
#define RAG_FAULT_BM25_LIST     4u  /* BM25 scan: a tile selected more than k documents       */
#define RAG_FAULT_BM25_POSTING  8u  /* BM25 scan: a posting's row lies outside its tile       */
#define RAG_FAULT_BM25_TERM    16u  /* BM25: query term id >= vocabulary size                 */
#define RAG_FAULT_FUSE_ROW     32u  /* fusion: row id outside the canon table                 */
#define RAG_FAULT_CE_PAIR      64u  /* pair packing: query, chunk row or slot out of range    */
#define RAG_FAULT_SEGMENT     128u  /* segmented top-k: offsets not ascending / out of range   */
#define RAG_FAULT_TOKEN       256u  /* embedding lookup: token, position or type id out of range */
int rag_fault_flags(uint32_t* out_flags, int32_t clear);

/* ---- dense exact kNN ----------------------------------------------------------
 * Replaces the embedded Qdrant store behind VectorIndexer.index_chunks
 * (reference src/indexing/vector_indexer.py:105-163: one fp32 vector per chunk,
 * id = row index) and the brute-force query_points call of VectorIndexer.search
 * (reference src/indexing/vector_indexer.py:202-209).                             */

/* rows: [n_rows x dim] fp32 row-major; dim must be a multiple of 4.              */
int rag_dense_create(const float* rows, int64_t n_rows, int32_t dim, int32_t metric,
                     int32_t mem_kind, int64_t row_base, rag_dense_t* out);
int rag_dense_destroy(rag_dense_t h);
int64_t rag_dense_rows(rag_dense_t h);
/* enable != 0: keep a second image of the rows (same size plus 256 rows) in which every value is split into bf16
 * high and low parts, and run searches as hi.hi + hi.lo + lo.hi on the bf16 matrix cores with fp32
 * accumulation (queries are split per search).  Those scores differ from the fp32 products by at most
 * 2^-16 |row||query| (1e-7 to 1e-6 for unit vectors) and only choose the rows: the scores returned
 * are plain fp32 scores of the chosen rows, in order.  A fifth of the matrix time.  dim must be a
 * multiple of 32.
 * enable == 0 drops the image: searches use exact fp32 FMA chains again (the default).          */
int rag_dense_enable_split(rag_dense_t h, int32_t enable);
/* Measurement aid (no counterpart in the reference): while enabled, every search records a
 * HIP event pair on the launch stream around its scan (the threshold pre-pass over a sample
 * of rows, when one runs, and the scan kernel; not the final merge), up to 256 searches;
 * rag_dense_profile_read waits for them and returns the durations in ms.             */
int rag_dense_profile(rag_dense_t h, int32_t enable);
int rag_dense_profile_read(rag_dense_t h, float* out_ms, int32_t max_n, int32_t* out_n);

/* Grow the handle's work buffers now to what searches of up to max_queries queries and max_k results need, so that
 * later rag_dense_search_dev calls within those bounds never call hipMalloc / hipFree (a serving loop, a second
 * stream, a graph capture).  No counterpart in the reference.                                          */
int rag_dense_reserve(rag_dense_t h, int32_t max_queries, int32_t max_k);
/* queries [n_queries x dim] fp32 (already normalised for cosine).  row_mask: NULL or
 * one byte per LOCAL row, 0 = excluded (the payload filter of
 * vector_indexer.py:185-196).  Outputs [n_queries x k]; out_count[q] = entries valid. */
int rag_dense_search_dev(rag_dense_t h, const float* d_queries, int32_t n_queries, int32_t k,
                         const uint8_t* d_row_mask, float* d_out_score, int64_t* d_out_row,
                         int32_t* d_out_count, void* stream);
int rag_dense_search(rag_dense_t h, const float* queries, int32_t n_queries, int32_t k,
                     const uint8_t* row_mask, float* out_score, int64_t* out_row,
                     int32_t* out_count);

/* Payload filter on the device: the exact-match filter of VectorIndexer.search (reference
 * src/indexing/vector_indexer.py:185-196: every key == value must hold) over per-row categorical
 * codes.  d_codes[n_fields][n_rows] int32 (one column per filterable key; the host assigns a code
 * per distinct value, a negative code to rows that lack the key); want[n_fields] on the host: the
 * code a row must carry, RAG_FILTER_ANY for an unconstrained key, RAG_FILTER_NONE when the value
 * occurs in no row (then no row passes).  Writes d_mask[n_rows] (1 = keep), the row_mask argument
 * of rag_dense_search[_dev].  At most 8 fields.                                          */
#define RAG_FILTER_ANY (-1)
#define RAG_FILTER_NONE (-2)
int rag_filter_mask_dev(const int32_t* d_codes, int32_t n_fields, int64_t n_rows, const int32_t* want,
                        uint8_t* d_mask, void* stream);
/* Exact fp32 scores of given candidate rows (GLOBAL ids, -1 = skip -> score 0):
 * out[q*m+j] = score(query q, row cand[q*m+j]).  Serves SimpleSimilarityReranker
 * (reference src/reranker/rerank.py:60-68) from the resident embeddings.          */
int rag_dense_rescore_dev(rag_dense_t h, const float* d_queries, int32_t n_queries,
                          const int64_t* d_cand_rows, int32_t m, float* d_out_score, void* stream);
int rag_dense_rescore(rag_dense_t h, const float* queries, int32_t n_queries,
                      const int64_t* cand_rows, int32_t m, float* out_score);

/* ---- BM25 ---------------------------------------------------------------------
 * Replaces rank_bm25.BM25Okapi.get_scores + np.argsort behind BM25Indexer.search
 * (reference src/indexing/bm25_indexer.py:127-128).  The index is term-major CSR:
 * indptr[vocab+1], rows[nnz] ascending inside a term, impact[nnz] fp64 where
 * impact = tf*(k1+1) / (tf + k1*(1 - b + b*doc_len/avgdl)), and idf[vocab] fp64
 * (epsilon floor already applied).  score(doc) = sum over query tokens, in query
 * order, of idf[t]*impact(t,doc), accumulated in fp64 without contraction.         */
int rag_bm25_create(const int64_t* indptr, const uint32_t* rows, const double* impact,
                    const double* idf, int64_t n_docs, int64_t vocab, int32_t mem_kind,
                    int64_t row_base, rag_bm25_t* out);
int rag_bm25_destroy(rag_bm25_t h);
/* Host side of the index build, native: tokenizer and vocabulary in one pass over the corpus bytes
 * (BM25Indexer.tokenize and the tokenising loop of BM25Indexer.index_chunks, reference
 * src/indexing/bm25_indexer.py:50-78 and :92-100).  text holds the documents back to back (UTF-8),
 * doc_off[n_docs+1] their byte offsets.  A document with pretokenized[d] == 0 (or pretokenized NULL) must be
 * ASCII and is tokenised here: lower-cased, maximal runs of [a-z0-9_] (what re.findall(r'\b\w+\b') yields on
 * ASCII text), tokens of one character dropped unless the character is one of short_keys[n_short_keys] (the
 * one-character lower-cased abbreviation keys) and occurs in the document.  A document with pretokenized[d] != 0
 * is taken as already tokenised by the caller, its tokens separated by '\n' (how documents outside ASCII come in:
 * Unicode word classes and case folding are left to the caller).  Vocabulary ids are given in order of first
 * appearance over the corpus.  No GPU work.  Fetch the result with the two calls below (NULL skips an output):
 * term_ids[n_tokens] document after document (the d_term of rag_bm25_build_dev), doc_len[n_docs],
 * vocab_bytes / vocab_off[n_vocab+1] the vocabulary's tokens back to back.                          */
int rag_bm25_tokenize_create(const char* text, const int64_t* doc_off, const uint8_t* pretokenized, int64_t n_docs,
                             const char* short_keys, int32_t n_short_keys, rag_tokenized_t* out);
int rag_bm25_tokenize_sizes(rag_tokenized_t h, int64_t* n_tokens, int64_t* n_vocab, int64_t* vocab_bytes);
int rag_bm25_tokenize_fetch(rag_tokenized_t h, int32_t* term_ids, int64_t* doc_len, char* vocab_bytes, int64_t* vocab_off);
void rag_bm25_tokenize_destroy(rag_tokenized_t h);
/* Index build on the device (the counting that rank_bm25's BM25Okapi constructor does behind
 * BM25Indexer.index_chunks, reference src/indexing/bm25_indexer.py:92-106): tokenised documents
 * -> term-major CSR.  d_term[n_tokens] holds the term ids document after document (negative =
 * token to ignore), d_doc_off[n_docs+1] the token offsets.  Writes d_indptr[vocab+1] and, for
 * every (term, document) pair that occurs, d_rows / d_tf (rows ascending inside a term);
 * *out_nnz = number of pairs, which must fit `capacity` (n_tokens always does; RAG_ELIMIT with
 * *out_nnz set otherwise).  Integer work: identical to counting on the host.  Returns after the
 * stream has finished (it allocates and frees its scratch).                              */
int rag_bm25_build_dev(const int32_t* d_term, int64_t n_tokens, const int64_t* d_doc_off,
                       int64_t n_docs, int64_t vocab, int64_t capacity, int64_t* d_indptr,
                       uint32_t* d_rows, uint32_t* d_tf, int64_t* out_nnz, void* stream);
/* impact[i] = tf (k1+1) / (tf + k1 (1 - b + b doc_len[row] / avgdl)) in fp64, every operation
 * rounded on its own in the operand order of rank_bm25's get_scores expression, so a posting's
 * impact is the value the reference multiplies by idf.                                   */
int rag_bm25_impacts_dev(const uint32_t* d_rows, const uint32_t* d_tf, int64_t nnz,
                         const int64_t* d_doc_len, int64_t n_docs, double avgdl, double k1, double b,
                         double* d_impact, void* stream);
int64_t rag_bm25_docs(rag_bm25_t h);

/* qterms: term ids of all queries back to back (-1 = out of vocabulary, ignored);
 * qoff[n_queries+1] delimits each query.  Returns the k best documents per query,
 * zero-score documents included when fewer than k documents match (as the
 * reference's argsort window does).                                               */
int rag_bm25_search_dev(rag_bm25_t h, const int32_t* d_qterms, const int32_t* d_qoff,
                        int32_t n_queries, int32_t n_terms_total, int32_t k,
                        double* d_out_score, int64_t* d_out_row, int32_t* d_out_count, void* stream);
/* The same for BM25: work buffers for up to max_queries queries with max_terms_total query tokens in all and
 * max_k results, so that rag_bm25_search_dev within those bounds allocates nothing.                  */
int rag_bm25_reserve(rag_bm25_t h, int32_t max_queries, int32_t max_terms_total, int32_t max_k);
int rag_bm25_search(rag_bm25_t h, const int32_t* qterms, const int32_t* qoff, int32_t n_queries,
                    int32_t k, double* out_score, int64_t* out_row, int32_t* out_count);
/* rag_bm25_search_dev for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same
 * arguments, same kernels and bit-identical results, but the work buffers (posting offsets per query term
 * and document tile, per-query thresholds, candidate lists) come from a static device array of
 * rag_bm25_search_static_capacity() bytes (32 MiB) that every GPU the library runs on holds until the
 * process ends, instead of the handle's grow-only buffers: the call never reaches hipMalloc or hipFree.
 * Precondition: rag_init(device) has run for the index's device, which must be the current device (rag_init
 * looks up that device's address of the array and allocates its fault word); otherwise RAG_EINVAL.
 * A search that needs more than the array holds returns RAG_ELIMIT and leaves the outputs untouched; the
 * need is about  n_queries * ceil(n_docs / 2048) * k * 12  +  n_terms_total * (ceil(n_docs / 2048) + 1) * 4
 * bytes (1M documents, k = 100: up to 55 queries per call).  All calls on one GPU share that GPU's
 * array: calls issued on different streams or threads are chained on the device (each waits for the one
 * before it through an event), so they do not race; a call on a capturing stream is not chained, and graph
 * replays that overlap other searches on the same GPU race unless the caller serialises them.           */
int64_t rag_bm25_search_static_capacity(void);
int rag_bm25_search_static_dev(rag_bm25_t h, const int32_t* d_qterms, const int32_t* d_qoff,
                               int32_t n_queries, int32_t n_terms_total, int32_t k,
                               double* d_out_score, int64_t* d_out_row, int32_t* d_out_count, void* stream);
/* Dense fp64 score vector of ONE query over all local documents (get_scores itself);
 * out_scores[n_docs] is host memory.  For parity checks and small corpora.         */
int rag_bm25_scores(rag_bm25_t h, const int32_t* qterms, int32_t n_terms, double* out_scores);

/* ---- top-k merge (multi-GPU shard merge) -----------------------------------------
 * Merges per-shard best-first lists gathered from all GPUs into one top-k_out list per
 * query with the same ordering rule.  Inputs [n_queries x n_lists x k_in]; counts
 * [n_queries x n_lists] or NULL (then rows < 0 mark unused slots).                 */
int rag_merge_topk_f32_dev(const float* d_score, const int64_t* d_row, const int32_t* d_count,
                           int32_t n_queries, int32_t n_lists, int32_t k_in, int32_t k_out,
                           int32_t smaller_is_better, float* d_out_score, int64_t* d_out_row,
                           int32_t* d_out_count, void* stream);
int rag_merge_topk_f64_dev(const double* d_score, const int64_t* d_row, const int32_t* d_count,
                           int32_t n_queries, int32_t n_lists, int32_t k_in, int32_t k_out,
                           double* d_out_score, int64_t* d_out_row, int32_t* d_out_count,
                           void* stream);

/* ---- rank fusion ---------------------------------------------------------------
 * Replaces HybridIndexer._reciprocal_rank_fusion (reference
 * src/indexing/hybrid_indexer.py:112-156) and _weighted_fusion (:158-218).
 * Per query: a vector list (rows [n_queries x vec_stride], vec_count[q] valid) and a
 * BM25 list.  canon: NULL, or canon[global row] = the first global row that carries the
 * same chunk_id (the reference fuses on the chunk_id string, which need not be unique);
 * n_canon = entries of that table (rows outside it fuse on their own id and are flagged).
 * Scores of both lists are fp64 (widen fp32 dense scores exactly before the call).
 * Outputs [n_queries x top_k]: out_pos = position of the payload-supplying entry in the
 * concatenation vector-list + BM25-list, out_row = its global row, out_score fp64.   */
int rag_fuse_rrf_dev(const int64_t* d_vec_row, const int32_t* d_vec_count, int32_t vec_stride,
                     const int64_t* d_bm_row, const int32_t* d_bm_count, int32_t bm_stride,
                     const uint32_t* d_canon, int64_t n_canon, int32_t n_queries, int32_t top_k,
                     double k_const, int32_t* d_out_pos, int64_t* d_out_row, double* d_out_score,
                     int32_t* d_out_count, void* stream);
int rag_fuse_weighted_dev(const int64_t* d_vec_row, const double* d_vec_score,
                          const int32_t* d_vec_count, int32_t vec_stride,
                          const int64_t* d_bm_row, const double* d_bm_score,
                          const int32_t* d_bm_count, int32_t bm_stride,
                          const uint32_t* d_canon, int64_t n_canon, int32_t n_queries, int32_t top_k,
                          double vector_weight, double bm25_weight,
                          int32_t* d_out_pos, int64_t* d_out_row, double* d_out_score,
                          int32_t* d_out_count, void* stream);
/* host-pointer forms (canon is host memory of n_canon entries, or NULL) */
int rag_fuse_rrf(const int64_t* vec_row, const int32_t* vec_count, int32_t vec_stride,
                 const int64_t* bm_row, const int32_t* bm_count, int32_t bm_stride,
                 const uint32_t* canon, int64_t n_canon, int32_t n_queries, int32_t top_k,
                 double k_const, int32_t* out_pos, int64_t* out_row, double* out_score,
                 int32_t* out_count);
int rag_fuse_weighted(const int64_t* vec_row, const double* vec_score, const int32_t* vec_count,
                      int32_t vec_stride, const int64_t* bm_row, const double* bm_score,
                      const int32_t* bm_count, int32_t bm_stride, const uint32_t* canon,
                      int64_t n_canon, int32_t n_queries, int32_t top_k, double vector_weight,
                      double bm25_weight, int32_t* out_pos, int64_t* out_row, double* out_score,
                      int32_t* out_count);

/* ---- rerank ---------------------------------------------------------------------
 * Replaces the stable descending sort + cut of BaseReranker.rerank (reference
 * src/reranker/rerank.py:42-43) for a batch of pools: scores of pool q are
 * d_score[seg_off[q] .. seg_off[q+1]) of the n_scores given; out_idx [n_queries x top_k]
 * holds positions inside the pool, best first, ties in pool order.                  */
int rag_topk_segmented_dev(const float* d_score, int32_t n_scores, const int32_t* d_seg_off,
                           int32_t n_queries, int32_t top_k, int32_t* d_out_idx,
                           int32_t* d_out_count, void* stream);
int rag_topk_segmented(const float* score, const int32_t* seg_off, int32_t n_queries,
                       int32_t top_k, int32_t* out_idx, int32_t* out_count);

/* Cross-encoder candidate batching: builds the padded token batch that
 * CrossEncoder.predict builds on the host (reference src/reranker/rerank.py:79-84):
 * pair p = (query pair_query[p], chunk row pair_row[p]) becomes
 * [CLS] query tokens [SEP] chunk tokens [SEP], the chunk part truncated so the pair
 * fits seq_len, written to row pair_slot[p] of the [n_slots x seq_len] outputs
 * (input ids, token type ids, attention mask), the rest padded with pad_id/0/0.
 * Token stores are CSR: tok[off[i] .. off[i+1]) for n_queries queries and n_chunks chunks
 * (chunk i is global row chunk_row_base + i).  pair_row < 0 yields an all-pad row; so does a
 * pair whose query or row lies outside the stores (flagged).  pair_slot NULL = slot p.   */
int rag_ce_pack_dev(const int32_t* d_query_tok, const int64_t* d_query_off, int32_t n_queries,
                    const int32_t* d_chunk_tok, const int64_t* d_chunk_off, int64_t n_chunks,
                    int64_t chunk_row_base,
                    const int32_t* d_pair_query, const int64_t* d_pair_row,
                    const int32_t* d_pair_slot, int32_t n_pairs, int32_t n_slots, int32_t seq_len,
                    int32_t cls_id, int32_t sep_id, int32_t pad_id,
                    int64_t* d_input_ids, int64_t* d_token_type_ids, int64_t* d_attention_mask,
                    void* stream);
/* The same pairs without padding: pair p occupies tokens [cu[p], cu[p+1]) of the
 * [total_tokens] outputs (input ids, token type ids, positions 0..len-1; int32), where
 * cu is the running sum of rag_ce_pair_len_dev's lengths for max_len (pairs with
 * pair_row < 0 have length 0).  A pair whose slice disagrees with its length is skipped
 * and flagged.                                                                         */
int rag_ce_pack_packed_dev(const int32_t* d_query_tok, const int64_t* d_query_off, int32_t n_queries,
                           const int32_t* d_chunk_tok, const int64_t* d_chunk_off, int64_t n_chunks,
                           int64_t chunk_row_base, const int32_t* d_pair_query,
                           const int64_t* d_pair_row, const int32_t* d_cu, int32_t n_pairs,
                           int64_t total_tokens, int32_t max_len, int32_t cls_id, int32_t sep_id,
                           int32_t* d_input_ids, int32_t* d_token_type_ids, int32_t* d_positions,
                           void* stream);
/* Length of each pair as rag_ce_pack_dev would write it (before padding, capped at
 * max_len); used to sort pairs into length buckets.                                */
int rag_ce_pair_len_dev(const int64_t* d_query_off, int32_t n_queries, const int64_t* d_chunk_off,
                        int64_t n_chunks, int64_t chunk_row_base,
                        const int32_t* d_pair_query, const int64_t* d_pair_row, int32_t n_pairs,
                        int32_t max_len, int32_t* d_out_len, void* stream);

/* ---- transformer encoder pieces ---------------------------------------------------
 * The two BERT forwards of the path (SentenceTransformer.encode behind
 * MedicalEmbedder.encode, reference src/indexing/embedder.py:65-71; CrossEncoder.predict
 * behind CrossEncoderReranker.score_pairs, reference src/reranker/rerank.py:79-84) keep
 * their GEMMs in hipBLASLt; these entry points are what runs between the GEMMs.        */
#define RAG_DTYPE_F32   0
#define RAG_DTYPE_BF16  1
#define RAG_DTYPE_F16   2   /* float16: the half type of split-precision operands, and the element type of the
                             * 16-bit instances of the attention / linear384 / mlp / LayerNorm / embedding kernels */
#define RAG_ENC_MAX_SEQ 512   /* longest sequence of rag_enc_attention_dev              */

/* Self-attention of n_seq padded sequences: qkv is the packed projection output
 * [n_seq][seq_len][3][n_heads][head_dim] (query, key, value), lens[i] = real tokens of
 * sequence i (keys at or beyond it are masked, as BERT's additive attention mask does);
 * out [n_seq][seq_len][n_heads*head_dim] = softmax(Q K^T / sqrt(head_dim)) V per head.
 * Rows of padding positions are written (finite), rows of 32-row blocks that hold only
 * padding are zero.  dtype bf16 or fp16 (MFMA kernel, fp32 softmax) or fp32 (plain fp32
 * arithmetic); head_dim 32 or 64.                                                        */
int rag_enc_attention_dev(const void* d_qkv, const int32_t* d_lens, int32_t n_seq, int32_t seq_len,
                          int32_t n_heads, int32_t head_dim, int32_t dtype, void* d_out, void* stream);
/* The same for a PACKED batch: no padding, sequence i owns token rows
 * [cu_seqlens[i], cu_seqlens[i+1]) of qkv [total_tokens][3][n_heads][head_dim] and of out;
 * max_seq_len bounds every sequence; a sequence longer than that or reaching beyond
 * total_tokens is skipped and flagged (RAG_FAULT_SEGMENT).                              */
int rag_enc_attention_packed_dev(const void* d_qkv, const int32_t* d_cu_seqlens, int32_t n_seq,
                                 int64_t total_tokens, int32_t max_seq_len, int32_t n_heads,
                                 int32_t head_dim, int32_t dtype, void* d_out, void* stream);
/* The packed form with only the FIRST token of every sequence as a query: out
 * [n_seq][n_heads*head_dim].  A sequence classifier reads only the [CLS] row of the last
 * layer, so that layer needs nothing else (bf16 only).                                  */
int rag_enc_attention_packed_first_dev(const void* d_qkv, const int32_t* d_cu_seqlens, int32_t n_seq,
                                       int64_t total_tokens, int32_t max_seq_len, int32_t n_heads,
                                       int32_t head_dim, int32_t dtype, void* d_out, void* stream);
/* fp32-equivalent self-attention of a PACKED batch on the bf16 matrix cores, for running the
 * cross-encoder forward at the reference's precision (CrossEncoder.predict computes in float32,
 * reference src/reranker/rerank.py:79-84).  qkv [total_tokens][3][n_heads][head_dim] fp32 is the
 * q|k|v projection WITHOUT its bias; bias [3 * n_heads * head_dim] fp32 (or NULL) is added while the
 * operands are staged.  Every operand is split into bf16 high and low parts and every product taken
 * as hi.hi + lo.hi + hi.lo with fp32 accumulation (softmax in fp32).  half_dtype = RAG_DTYPE_BF16 or
 * RAG_DTYPE_F16 is the type of the halves, of the internal ones and of the output's (see
 * rag_enc_split_bf16x3_dev).  out [rows][3 * H], H = n_heads * head_dim: row = [hi | hi | lo] of the attention
 * output, the left operand of a split-precision GEMM (see rag_enc_split_bf16x3_dev); rows =
 * total_tokens, or n_seq with first_only != 0 (only the first token of every sequence is a query:
 * the last layer of a sequence classifier).  head_dim 32 or 64; max_seq_len bounds every sequence
 * (head_dim 32: up to 512 tokens; 64: up to 288 by LDS size).                                  */
int rag_enc_attention_split_dev(const float* d_qkv, const float* d_bias, const int32_t* d_cu_seqlens,
                                int32_t n_seq, int64_t total_tokens, int32_t max_seq_len, int32_t n_heads,
                                int32_t head_dim, int32_t first_only, int32_t half_dtype, void* d_out, void* stream);
/* The same with out [rows][2 * H]: row = [hi | lo], a third less to write.  For a consumer that reads the two
 * halves where they lie (rag_enc_linear_ln_split_dev with in_is_split == 2) rather than a library GEMM over
 * [hi | hi | lo].                                                                                  */
int rag_enc_attention_split_hl_dev(const float* d_qkv, const float* d_bias, const int32_t* d_cu_seqlens,
                                   int32_t n_seq, int64_t total_tokens, int32_t max_seq_len, int32_t n_heads,
                                   int32_t head_dim, int32_t first_only, int32_t half_dtype, void* d_out, void* stream);
/* Sentence embedding of a packed fp32 batch: out[i] = mean of x over the tokens of
 * sequence i (the mean pooling over the attention mask of the sentence-transformers
 * pipeline behind reference src/indexing/embedder.py:65-71), L2-normalised when
 * `normalize` is non-zero.  x [total_tokens][hidden], out [n_seq][hidden].             */
int rag_enc_mean_pool_dev(const float* d_x, const int32_t* d_cu_seqlens, int32_t n_seq,
                          int64_t total_tokens, int32_t hidden, int32_t normalize, float* d_out,
                          void* stream);
/* Feed-forward block of a BERT layer in one kernel (bf16 or fp16 by dtype, fp32 accumulation):
 *   y[m] = LayerNorm(x[m] + GELU(x[m] w1^T + b1) w2^T + b2) * gamma + beta
 * x, y [m][hidden]; w1 [intermediate][hidden], w2 [hidden][intermediate] as nn.Linear stores
 * them (the BertIntermediate / BertOutput sub-layers of the cross-encoder forward behind
 * reference src/reranker/rerank.py:79-84).  The [m][intermediate] activation stays on chip.
 * hidden == 384, intermediate a multiple of 32 up to 2048 (b1 and rings of three weight chunks per
 * matrix share the LDS); y must not alias x.  GELU is the erf form,
 * evaluated to 4e-5 absolute.                                                            */
int rag_enc_mlp_dev(const void* d_x, const void* d_w1, const void* d_b1, const void* d_w2,
                    const void* d_b2, const void* d_gamma, const void* d_beta, float eps, int64_t m,
                    int32_t hidden, int32_t intermediate, int32_t dtype, void* d_y, void* stream);
/* The rest of a BERT layer after self-attention in one kernel (bf16 or fp16, fp32 accumulation):
 *   x1   = LayerNorm(x[m] + ctx[m] wo^T + bo) * gamma1 + beta1      (BertSelfOutput)
 *   y[m] = LayerNorm(x1 + GELU(x1 w1^T + b1) w2^T + b2) * gamma2 + beta2
 * ctx = attention output, x = the layer's input, both [m][hidden]; wo [hidden][hidden]; the other
 * arguments as in rag_enc_mlp_dev (same reference path, src/reranker/rerank.py:79-84).  x1 is kept
 * on chip, rounded to the 16-bit type as the layer-by-layer forward stores it.  y may alias x or ctx only if
 * the caller no longer needs them: rows are read before they are written by the same workgroup. */
int rag_enc_attn_out_mlp_dev(const void* d_ctx, const void* d_x, const void* d_wo, const void* d_bo,
                             const void* d_gamma1, const void* d_beta1, const void* d_w1,
                             const void* d_b1, const void* d_w2, const void* d_b2,
                             const void* d_gamma2, const void* d_beta2, float eps, int64_t m,
                             int32_t hidden, int32_t intermediate, int32_t dtype, void* d_y,
                             void* stream);
/* The same block with only the residual + LayerNorm in front of it:
 *   x1 = LayerNorm(x + proj) * gamma1 + beta1,   y = LayerNorm(x1 + W2 GELU(W1 x1 + b1) + b2) * gamma2 + beta2
 * where proj is the finished attention-output projection (bias included, e.g. from
 * rag_enc_linear384_dev): BertSelfOutput's LayerNorm and BertIntermediate/BertOutput of a layer
 * (transformers, reached through CrossEncoder.predict, reference src/reranker/rerank.py:83) without
 * the separate pass over the activations for the first LayerNorm.  Same limits as rag_enc_mlp_dev. */
int rag_enc_norm_mlp_dev(const void* d_proj, const void* d_x, const void* d_gamma1, const void* d_beta1,
                         const void* d_w1, const void* d_b1, const void* d_w2, const void* d_b2,
                         const void* d_gamma2, const void* d_beta2, float eps, int64_t m, int32_t hidden,
                         int32_t intermediate, int32_t dtype, void* d_y, void* stream);
/* y[m][n] = sum_k x[m][k] w[n][k] + bias[n] for 384-wide inputs (bf16 or fp16, fp32 accumulation): the
 * q|k|v and attention-output projections of the cross-encoder layers (same reference path as
 * rag_enc_mlp_dev).  x [m][384], w [n][384] as nn.Linear stores it, n a multiple of 32; y must not
 * alias x.  With d_residual != NULL (n == 384): y = LayerNorm(that + residual[m]) * gamma + beta,
 * the sum rounded to bf16 before the statistics as a separate residual + LayerNorm pass over a
 * bf16 GEMM output would see it; y may alias residual.                                  */
int rag_enc_linear384_dev(const void* d_x, const void* d_w, const void* d_bias, int64_t m, int32_t n,
                          const void* d_residual, const void* d_gamma, const void* d_beta, float eps,
                          int32_t dtype, void* d_y, void* stream);
/* q|k|v projection and self-attention of a packed batch in ONE kernel (bf16 or fp16, hidden 384, 12 heads of 32):
 *   out[t] = concat_h softmax(Q_h K_h^T / sqrt(32)) V_h   with   [Q | K | V] = x W^T + bias
 * — BertSelfAttention of the cross-encoder layers (transformers, reached through CrossEncoder.predict, reference
 * src/reranker/rerank.py:79-84) without the 2304 bytes a token that rag_enc_linear384_dev would write and
 * rag_enc_attention_packed_dev read back.  x, out [total_tokens][384]; w [1152][384] and bias [1152] as nn.Linear stores
 * the concatenated projection (heads inside each of q, k, v); cu_seqlens [n_seq + 1] as in rag_enc_attention_packed_dev.
 * The batch is processed in TILES of whole, consecutive sequences with at most 256 tokens each: d_tile_seq[t] = first
 * sequence of tile t, n_tiles + 1 entries on the device, the last one n_seq (or the number of leading sequences that
 * have tokens).  A tile that is not one (more than 256 tokens, sequences out of order or beyond the batch) is skipped
 * and flagged (RAG_FAULT_SEGMENT).  out must not alias x.                                                        */
int rag_enc_qkv_attention_dev(const void* d_x, const void* d_w, const void* d_bias, const int32_t* d_cu_seqlens,
                              int32_t n_seq, int64_t total_tokens, const int32_t* d_tile_seq, int32_t n_tiles,
                              int32_t n_heads, int32_t head_dim, int32_t dtype, void* d_out, void* stream);
/* The same with only the FIRST token of every sequence as a query (the [CLS] row, all the pooler behind the last layer
 * reads; as rag_enc_attention_packed_first_dev): out [n_seq][384], row = sequence.  The tile list must cover every
 * sequence (d_tile_seq[n_tiles] == n_seq).                                                                        */
int rag_enc_qkv_attention_first_dev(const void* d_x, const void* d_w, const void* d_bias, const int32_t* d_cu_seqlens,
                                    int32_t n_seq, int64_t total_tokens, const int32_t* d_tile_seq, int32_t n_tiles,
                                    int32_t n_heads, int32_t head_dim, int32_t dtype, void* d_out, void* stream);
/* Host helpers for the tile list of rag_enc_qkv_attention_dev (all arrays on the HOST; no counterpart in the reference,
 * whose CrossEncoder.predict pads mini-batches of 32 instead).  _tiles cuts the sequences as they come: h_cu [n_seq + 1]
 * token offsets -> h_tile_seq (room for n_seq + 1 entries), greedy, every sequence 1..256 tokens (RAG_ELIMIT otherwise).
 * _pack chooses the ORDER as well: h_len [n] tokens of each sequence -> h_order [n] (the sequence that comes j-th) and
 * h_tile_seq over that order; a tile takes the longest remaining sequence, then repeatedly the longest that still fits;
 * sequences without tokens come last, in no tile; RAG_ELIMIT if one has more than 256.                         */
int rag_enc_qkv_attention_tiles(const int64_t* h_cu, int32_t n_seq, int32_t* h_tile_seq, int32_t* n_tiles);
int rag_enc_qkv_attention_pack(const int32_t* h_len, int32_t n, int64_t* h_order, int32_t* h_tile_seq, int32_t* n_tiles);
/* Measurement aid (no counterpart in the reference): while enabled, every launch of the
 * feed-forward kernel (rag_enc_mlp_dev, rag_enc_attn_out_mlp_dev) records a HIP event pair on
 * its stream, up to 1024 launches; rag_enc_mlp_profile_read waits for them and returns their
 * durations in ms and row counts.  One state per process.                               */
int rag_enc_mlp_profile(int32_t enable);
int rag_enc_mlp_profile_read(float* out_ms, int64_t* out_rows, int32_t max_n, int32_t* out_n);
/* x := GELU(x) (exact erf form, BERT's "gelu"), in place over n elements (n % 8 == 0).  */
int rag_enc_gelu_dev(void* d_x, int64_t n, int32_t dtype, void* stream);
/* Split-precision operand for running an fp32 GEMM on the bf16 matrix cores: each fp32
 * row x[r][0..k) becomes the bf16 row [hi | hi | lo] of 3k elements (hi = bf16(x),
 * lo = bf16(x - hi)).  Multiplied (fp32 accumulation) by weights laid out as
 * [w_hi ; w_lo ; w_hi] along k it yields x_hi w_hi + x_hi w_lo + x_lo w_hi, i.e. the fp32
 * product to about 2^-16 relative per term (the dropped x_lo w_lo term).
 * half_dtype selects the 16-bit type of the halves (here and in the entry points below that write such
 * rows): RAG_DTYPE_BF16 as described, or RAG_DTYPE_F16: 11 significant bits per half instead of 8, so
 * the three products give the fp32 product to fp32 round-off (measured on a 4096 x 384 x 1536 GEMM of unit
 * normal inputs and 0.03-normal weights: mean error 3.0e-7 against 1.5e-7 for an fp32 GEMM and 2.1e-6 for
 * bf16 halves); fp16 subnormals are honoured by the matrix cores, values must stay below 65504.       */
int rag_enc_split_bf16x3_dev(const float* d_x, int64_t rows, int32_t k, int32_t half_dtype, void* d_out, void* stream);
/* The same split of act(y + bias) (bias [k] fp32; gelu != 0: exact erf GELU, else identity): what
 * stands between the two split-precision GEMMs of BERT's feed-forward block
 * (transformers BertIntermediate, reached through SentenceTransformer.encode,
 * reference src/indexing/embedder.py:65-71).                                              */
int rag_enc_bias_act_split_bf16x3_dev(const float* d_y, const float* d_bias, int32_t gelu, int64_t rows,
                                      int32_t k, int32_t half_dtype, void* d_out, void* stream);
/* out[r] = LayerNorm((y[r] + bias) + residual[r]) * gamma + beta in fp32 (BertSelfOutput /
 * BertOutput behind reference src/indexing/embedder.py:65-71), and, when d_out_split is not
 * NULL, the bf16 row [hi | hi | lo] of out[r] (3 * hidden elements) for the split-precision
 * GEMM that reads it next.  hidden a multiple of 8 up to 2048; out may alias y or residual. */
int rag_enc_bias_add_layernorm_split_dev(const float* d_y, const float* d_bias, const float* d_residual,
                                         const float* d_gamma, const float* d_beta, float eps, int64_t rows,
                                         int32_t hidden, int32_t half_dtype, float* d_out, void* d_out_split,
                                         void* stream);
/* Dense layer + residual + LayerNorm of a 384-wide BERT layer at float32 precision on the 16-bit matrix cores
 * (the BertOutput and BertSelfOutput sub-layers of the float32 cross-encoder forward behind
 * CrossEncoder.predict, reference src/reranker/rerank.py:79-84):
 *   y[m] = LayerNorm(residual[m] + a[m] W^T + b2) * gamma + beta,      W [n = 384][k], y fp32 [m][384]
 * where the left operand a is, with in_is_split == 0,  a = GELU(in + b1)  for in fp32 [m][k], the first
 * product of the feed-forward block without its bias (erf GELU, evaluated to fp32 round-off), or, with
 * in_is_split == 1, the operand given by in = fp16 [m][3k] rows [hi | hi | lo] (what rag_enc_split_bf16x3_dev
 * and rag_enc_attention_split_dev write with RAG_DTYPE_F16; b1 is ignored), or, with in_is_split == 2, by
 * in = fp16 [m][2k] rows [hi | lo] (rag_enc_attention_split_hl_dev).  W comes as its fp16 halves
 * wh = fp16(W), wl = fp16(W - wh), each 384 * k elements packed by rag_enc_linear_ln_split_pack_dev (once per
 * weight: the kernel streams them chunk by chunk).  The product is hi.hi + lo.hi + hi.lo with fp32
 * accumulation; bias, residual and LayerNorm are fp32.  d_y_split (may be NULL): fp16 [m][1152] rows
 * [hi | hi | lo] of y, the operand of the next split-precision GEMM.  k a multiple of 32 up to 2048;
 * outputs must not alias the input; residual may alias y.  d_y may be NULL when d_y_split is not (only the
 * split row is wanted: the fp32 row's stores are 9 % of a call with k = 384).                      */
int rag_enc_linear_ln_split_dev(const void* d_in, int32_t in_is_split, const float* d_b1, const void* d_wh,
                                const void* d_wl, const float* d_b2, const float* d_residual,
                                const float* d_gamma, const float* d_beta, float eps, int64_t m, int32_t k,
                                int32_t n, float* d_y, void* d_y_split, void* stream);
/* The same for a chain of these calls: the residual comes as d_residual_split = fp16 [m][1152] rows [hi | hi | lo]
 * (the d_y_split of the call that produced it; residual = hi + lo, which carries 22 of its 24 significant bits
 * - the same value the GEMMs between the calls are given), so that the producing call needs to write no fp32
 * row (d_y NULL there and here; d_y_split may then not be NULL).  d_y_split must not alias d_residual_split. */
int rag_enc_linear_ln_split_chain_dev(const void* d_in, int32_t in_is_split, const float* d_b1, const void* d_wh,
                                      const void* d_wl, const float* d_b2, const void* d_residual_split,
                                      const float* d_gamma, const float* d_beta, float eps, int64_t m, int32_t k,
                                      int32_t n, float* d_y, void* d_y_split, void* stream);
/* The weight operands of rag_enc_linear_ln_split_dev from W fp32 [n = 384][k] row-major (a Linear layer's weight as
 * transformers stores it): d_wh, d_wl each 384 * k fp16, where 16-byte piece 1536 c + 4 f + j holds columns
 * 32 c + 8 (j ^ ((f >> 2) & 3)) + 0..7 of row (f with bits 2 and 3 swapped): every 32-column chunk contiguous
 * and in the order of the kernel's LDS image.                                                        */
int rag_enc_linear_ln_split_pack_dev(const float* d_w, int32_t k, int32_t n, void* d_wh, void* d_wl, void* stream);
/* Measurement aid (no counterpart in the reference): HIP event pairs around the launches of
 * rag_enc_linear_ln_split_dev while enabled (up to 1024); the read returns durations in ms, row counts and k. */
int rag_enc_linear_ln_split_profile(int32_t enable);
int rag_enc_linear_ln_split_profile_read(float* out_ms, int64_t* out_rows, int32_t* out_k, int32_t max_n,
                                         int32_t* out_n);
/* BERT's input embedding: out[t] = LayerNorm(word[ids[t]] + position[pos[t]] +
 * type_emb[types[t]]) (types NULL = all 0), tables [vocab|max_pos|n_types][hidden];
 * ids outside a table are clamped and flagged (RAG_FAULT_TOKEN).                        */
int rag_enc_embed_layernorm_dev(const int32_t* d_ids, const int32_t* d_types, const int32_t* d_pos,
                                const void* d_word, const void* d_position, const void* d_type_emb,
                                const void* d_gamma, const void* d_beta, float eps, int64_t n_tokens,
                                int32_t hidden, int32_t vocab, int32_t max_pos, int32_t n_types,
                                int32_t dtype, void* d_out, void* stream);
/* The same for fp32 tables with the row also written as [hi | hi | lo] in `half_dtype` (out_split
 * [n_tokens][3 * hidden], what rag_enc_split_bf16x3_dev makes of `out`): the embedding output of
 * transformers' BertEmbeddings as the left operand of the first split-precision GEMM, in one pass
 * (reached through CrossEncoder.predict / SentenceTransformer.encode, reference
 * src/reranker/rerank.py:79-84, src/indexing/embedder.py:65-71).  d_out may be NULL: only the split row is
 * written (what a layer made of rag_enc_linear_ln_split_chain_dev calls reads).                    */
int rag_enc_embed_layernorm_split_dev(const int32_t* d_ids, const int32_t* d_types, const int32_t* d_pos,
                                      const float* d_word, const float* d_position, const float* d_type_emb,
                                      const float* d_gamma, const float* d_beta, float eps, int64_t n_tokens,
                                      int32_t hidden, int32_t vocab, int32_t max_pos, int32_t n_types,
                                      float* d_out, int32_t half_dtype, void* d_out_split, void* stream);
/* out[r] = LayerNorm(x[r] + residual[r]) * gamma + beta over `hidden` elements per row
 * (residual may be NULL); fp32 arithmetic, inputs/outputs bf16, fp16 or fp32 as `dtype` says;
 * hidden a multiple of 8 up to 2048.  out may alias x or residual.                     */
int rag_enc_add_layernorm_dev(const void* d_x, const void* d_residual, const void* d_gamma,
                              const void* d_beta, float eps, int64_t rows, int32_t hidden,
                              int32_t dtype, void* d_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RAG_HIP_H */
