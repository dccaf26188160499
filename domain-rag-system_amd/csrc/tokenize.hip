// Host side of the BM25 index build: tokenizer and vocabulary, native (no GPU work in this file).
//
// The reference tokenises every chunk in Python (BM25Indexer.tokenize, src/indexing/bm25_indexer.py:50-78:
// lower-case, re.findall(r'\b\w+\b'), drop one-character tokens unless they equal a lower-cased abbreviation key
// that occurs in the text) and rank_bm25 then walks the token lists once more to count them
// (BM25Indexer.index_chunks, bm25_indexer.py:92-106).  Here one pass over the corpus bytes produces what the device
// builder (rag_bm25_build_dev) consumes: term ids in document order, document lengths, and the vocabulary in
// first-appearance order.
//
// Unicode: Python's \w and str.lower() follow the Unicode tables.  This file does not restate them: a document
// made of ASCII bytes only is tokenised here ([A-Za-z0-9_] runs, A-Z lowered); any other document must come
// already tokenised by the caller (the Python tokenizer), as its tokens separated by '\n'.  Ids are given in
// corpus order across both kinds, so the result is the same as tokenising everything in Python.
#include "common.h"

#include <string>
#include <vector>

namespace {

struct TokBuild {
    std::vector<int32_t> term_ids;          // every kept token of every document, in order
    std::vector<int64_t> doc_len;           // tokens per document
    std::vector<char> vocab_bytes;          // the vocabulary's tokens back to back (UTF-8), first-appearance order
    std::vector<int64_t> vocab_off;         // [n_vocab + 1]
    // open-addressing hash table over the vocabulary: slot -> vocabulary id + 1 (0 = empty)
    std::vector<uint32_t> slots;
    std::vector<uint64_t> hashes;           // per vocabulary id
    size_t mask = 0;

    static uint64_t hash_bytes(const char* p, size_t n) {          // FNV-1a, 64 bit
        uint64_t h = 1469598103934665603ull;
        for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 1099511628211ull; }
        return h;
    }
    void grow() {
        const size_t cap = slots.empty() ? (1u << 16) : slots.size() * 2;
        slots.assign(cap, 0);
        mask = cap - 1;
        for (size_t id = 0; id < hashes.size(); ++id) {
            size_t s = hashes[id] & mask;
            while (slots[s]) s = (s + 1) & mask;
            slots[s] = (uint32_t)id + 1;
        }
    }
    int32_t intern(const char* p, size_t n) {
        if ((hashes.size() + 1) * 2 > slots.size()) grow();
        const uint64_t h = hash_bytes(p, n);
        size_t s = h & mask;
        while (slots[s]) {
            const uint32_t id = slots[s] - 1;
            if (hashes[id] == h) {
                const size_t a = (size_t)vocab_off[id], len = (size_t)(vocab_off[id + 1] - vocab_off[id]);
                if (len == n && memcmp(vocab_bytes.data() + a, p, n) == 0) return (int32_t)id;
            }
            s = (s + 1) & mask;
        }
        const uint32_t id = (uint32_t)hashes.size();
        hashes.push_back(h);
        vocab_bytes.insert(vocab_bytes.end(), p, p + n);
        vocab_off.push_back((int64_t)vocab_bytes.size());
        slots[s] = id + 1;
        return (int32_t)id;
    }
};

inline bool is_word(unsigned char c) {
    return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_';
}

}  // namespace

extern "C" int rag_bm25_tokenize_create(const char* text, const int64_t* doc_off, const uint8_t* pretokenized,
                                        int64_t n_docs, const char* short_keys, int32_t n_short_keys,
                                        rag_tokenized_t* out) {
    RAG_REQUIRE(out != nullptr, RAG_EINVAL, "rag_bm25_tokenize_create: null output handle");
    *out = nullptr;
    RAG_REQUIRE(n_docs >= 0 && doc_off != nullptr && (n_docs == 0 || text != nullptr), RAG_EINVAL,
                "rag_bm25_tokenize_create: null corpus");
    RAG_REQUIRE(n_short_keys >= 0 && (n_short_keys == 0 || short_keys != nullptr), RAG_EINVAL,
                "rag_bm25_tokenize_create: bad abbreviation key list");
    for (int64_t d = 0; d < n_docs; ++d)
        RAG_REQUIRE(doc_off[d + 1] >= doc_off[d] && doc_off[0] == 0, RAG_EINVAL,
                    "rag_bm25_tokenize_create: document offsets must start at 0 and not decrease (document %lld)", (long long)d);
    TokBuild* b = new TokBuild();
    b->vocab_off.push_back(0);
    b->doc_len.assign((size_t)n_docs, 0);
    std::string low;
    for (int64_t d = 0; d < n_docs; ++d) {
        const char* p = text + doc_off[d];
        const size_t n = (size_t)(doc_off[d + 1] - doc_off[d]);
        int64_t kept = 0;
        if (pretokenized != nullptr && pretokenized[d]) {
            // the caller's tokens, one per line ('\n' is not a word character, so no token contains it); an
            // empty document has no bytes
            size_t a = 0;
            while (a < n) {
                size_t e = a;
                while (e < n && p[e] != '\n') ++e;
                b->term_ids.push_back(b->intern(p + a, e - a));
                ++kept;
                a = e + 1;
            }
        } else {
            low.assign(p, n);
            bool rescued[256] = {};
            bool any_rescue = false;
            for (size_t i = 0; i < n; ++i) {
                const unsigned char c = (unsigned char)low[i];
                if (c >= 0x80) {
                    delete b;
                    rag_set_error("rag_bm25_tokenize_create: document %lld holds non-ASCII bytes and is not marked "
                                  "as tokenised by the caller", (long long)d);
                    return RAG_EINVAL;
                }
                if (c >= 'A' && c <= 'Z') low[i] = (char)(c + 32);
            }
            // one-character abbreviation keys that occur anywhere in the lower-cased text keep their token
            for (int32_t k = 0; k < n_short_keys; ++k) {
                const unsigned char key = (unsigned char)short_keys[k];
                if (key < 0x80 && memchr(low.data(), key, n) != nullptr) { rescued[key] = true; any_rescue = true; }
            }
            size_t i = 0;
            while (i < n) {
                if (!is_word((unsigned char)low[i])) { ++i; continue; }
                size_t e = i;
                while (e < n && is_word((unsigned char)low[e])) ++e;
                if (e - i >= 2 || (any_rescue && rescued[(unsigned char)low[i]])) {
                    b->term_ids.push_back(b->intern(low.data() + i, e - i));
                    ++kept;
                }
                i = e;
            }
        }
        b->doc_len[(size_t)d] = kept;
    }
    *out = reinterpret_cast<rag_tokenized_t>(b);
    return RAG_OK;
}

extern "C" int rag_bm25_tokenize_sizes(rag_tokenized_t h, int64_t* n_tokens, int64_t* n_vocab, int64_t* vocab_bytes) {
    RAG_REQUIRE(h != nullptr && n_tokens && n_vocab && vocab_bytes, RAG_EINVAL, "rag_bm25_tokenize_sizes: null argument");
    const TokBuild* b = reinterpret_cast<const TokBuild*>(h);
    *n_tokens = (int64_t)b->term_ids.size();
    *n_vocab = (int64_t)b->hashes.size();
    *vocab_bytes = (int64_t)b->vocab_bytes.size();
    return RAG_OK;
}

extern "C" int rag_bm25_tokenize_fetch(rag_tokenized_t h, int32_t* term_ids, int64_t* doc_len, char* vocab_bytes,
                                       int64_t* vocab_off) {
    RAG_REQUIRE(h != nullptr, RAG_EINVAL, "rag_bm25_tokenize_fetch: null handle");
    const TokBuild* b = reinterpret_cast<const TokBuild*>(h);
    if (term_ids && !b->term_ids.empty()) memcpy(term_ids, b->term_ids.data(), b->term_ids.size() * sizeof(int32_t));
    if (doc_len && !b->doc_len.empty()) memcpy(doc_len, b->doc_len.data(), b->doc_len.size() * sizeof(int64_t));
    if (vocab_bytes && !b->vocab_bytes.empty()) memcpy(vocab_bytes, b->vocab_bytes.data(), b->vocab_bytes.size());
    if (vocab_off) memcpy(vocab_off, b->vocab_off.data(), b->vocab_off.size() * sizeof(int64_t));
    return RAG_OK;
}

extern "C" void rag_bm25_tokenize_destroy(rag_tokenized_t h) {
    delete reinterpret_cast<TokBuild*>(h);
}
