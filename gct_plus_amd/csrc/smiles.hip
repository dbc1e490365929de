// Host-side SMILES tokeniser + numericaliser (no GPU work; lives in the same C ABI library).
// Reference: Utils/field.py:8-33 `moltokenize` -- regex
//   (\[[^\]]+]|Br?|Cl?|N|O|S|P|F|I|b|c|n|o|s|p|\(|\)|\.|=|#|-|\+|\\|\/|:|~|@|\?|>|\*|\$|\%[0-9]{2}|[0-9])
// applied with findall (unmatched characters are skipped); with add_sep the literal "<sep>"
// splits scaffold and molecule (field.py:25-33).  Restated as a single-pass scanner.
// Numericalisation = torchtext Field.process as used by Model/collate_fn.py:5-15,104-124:
// [<sos>] tokens [<eos>] then <pad> to the row width; unknown tokens -> <unk>.
#include <string.h>

#include "common.h"

namespace {

inline bool is_digit(char c) { return c >= '0' && c <= '9'; }

// length of the token starting at s[i] (0 = no token starts here, skip one character)
inline int token_len(const char* s, int i, int n, bool with_sep) {
  const char c = s[i];
  if (with_sep && c == '<' && i + 4 < n + 0 && strncmp(s + i, "<sep>", 5) == 0) return 5;
  switch (c) {
    case '[': {
      int j = i + 1;
      while (j < n && s[j] != ']') ++j;
      if (j < n && j > i + 1) return j - i + 1;   // at least one char inside the brackets
      return 0;
    }
    case 'B': return (i + 1 < n && s[i + 1] == 'r') ? 2 : 1;
    case 'C': return (i + 1 < n && s[i + 1] == 'l') ? 2 : 1;
    case 'N': case 'O': case 'S': case 'P': case 'F': case 'I':
    case 'b': case 'c': case 'n': case 'o': case 's': case 'p':
    case '(': case ')': case '.': case '=': case '#': case '-': case '+': case '\\': case '/':
    case ':': case '~': case '@': case '?': case '>': case '*': case '$':
      return 1;
    case '%': return (i + 2 < n && is_digit(s[i + 1]) && is_digit(s[i + 2])) ? 3 : 0;
    default: return is_digit(c) ? 1 : 0;
  }
}

}  // namespace

extern "C" int gct_smiles_tokenize(const char* smi, int with_sep, int32_t* tok_start,
                                   int32_t* tok_len, int max_tokens) {
  GCT_CHECK_ARG(smi && max_tokens >= 0, "smiles_tokenize: bad args");
  const int n = (int)strlen(smi);
  int cnt = 0;
  // field.py:25-33: with_sep and more than one "<sep>" => empty token list
  if (with_sep) {
    int seps = 0;
    for (const char* p = strstr(smi, "<sep>"); p; p = strstr(p + 5, "<sep>")) ++seps;
    if (seps > 1) return 0;
  }
  for (int i = 0; i < n;) {
    const int l = token_len(smi, i, n, with_sep != 0);
    if (l == 0) { ++i; continue; }
    if (cnt < max_tokens && tok_start && tok_len) {
      tok_start[cnt] = i;
      tok_len[cnt] = l;
    }
    ++cnt;
    i += l;
  }
  return cnt;
}

extern "C" int gct_smiles_encode_batch(const char* const* smiles, int n, int with_sep,
                                       const char* const* vocab, int vocab_size, int64_t unk_id,
                                       int64_t pad_id, int64_t sos_id, int64_t eos_id, int64_t* out,
                                       int64_t width, int32_t* lengths) {
  GCT_CHECK_ARG(smiles && vocab && out && n >= 0 && vocab_size > 0 && width > 0,
                "smiles_encode_batch: bad args");
  int vlen[4096];
  GCT_CHECK_ARG(vocab_size <= 4096, "smiles_encode_batch: vocabulary too large");
  for (int v = 0; v < vocab_size; ++v) vlen[v] = (int)strlen(vocab[v]);
  int maxlen = 0;
  for (int r = 0; r < n; ++r) {
    const char* s = smiles[r];
    const int sl = (int)strlen(s);
    int64_t* row = out + (int64_t)r * width;
    int64_t w = 0;
    if (sos_id >= 0) row[w++] = sos_id;
    bool empty = false;
    if (with_sep) {
      int seps = 0;
      for (const char* p = strstr(s, "<sep>"); p; p = strstr(p + 5, "<sep>")) ++seps;
      empty = seps > 1;
    }
    for (int i = 0; i < sl && !empty;) {
      const int l = token_len(s, i, sl, with_sep != 0);
      if (l == 0) { ++i; continue; }
      int64_t id = unk_id;
      for (int v = 0; v < vocab_size; ++v)
        if (vlen[v] == l && memcmp(vocab[v], s + i, l) == 0) { id = v; break; }
      if (w >= width) {
        gct_set_error("smiles_encode_batch: row %d needs more than %ld columns", r, (long)width);
        return GCT_ERR_ARG;
      }
      row[w++] = id;
      i += l;
    }
    if (eos_id >= 0) {
      if (w >= width) {
        gct_set_error("smiles_encode_batch: row %d needs more than %ld columns", r, (long)width);
        return GCT_ERR_ARG;
      }
      row[w++] = eos_id;
    }
    if (lengths) lengths[r] = (int32_t)w;
    if (w > maxlen) maxlen = (int)w;
    for (; w < width; ++w) row[w] = pad_id;
  }
  return maxlen;
}
