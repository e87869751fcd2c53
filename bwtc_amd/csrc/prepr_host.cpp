// See prepr_host.hpp.  Reference citations are file:line under /root/reference.
#include "prepr_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace bwtc {
namespace prepr {

// ---- Grammar ---------------------------------------------------------------------------------------
Grammar::Grammar() : new_rules_(0) {                               // Grammar.cpp:36-40
  std::memset(freq_, 0, sizeof freq_);
  std::memset(special_, 0, sizeof special_);
  std::memset(variable_, 0, sizeof variable_);
}

void Grammar::addRule(byte variable, byte first, byte second) {    // Grammar.cpp:42-51
  Rule r;
  r.variable = variable;
  r.large = false;
  r.rhs.push_back(first);
  r.rhs.push_back(second);
  rules_.push_back(r);
  ++freq_[first]; ++freq_[second]; ++freq_[variable];
  ++new_rules_;
}

// Special pairs are numbered so that the pairs over the first k special symbols are the numbers below k^2:
// (i, i) -> i^2, (i, k) with i < k -> k^2 + 1 + i, (k, i) with i < k -> k^2 + k + 1 + i  (Grammar.cpp:160-183).
uint32_t Grammar::ordinalOfPair(uint32_t first, uint32_t second) const {
  if (first == second) return first * second;
  if (first > second) return first * (first + 1) + second + 1;
  return second * second + first + 1;
}
uint16_t Grammar::specialPair(uint32_t ord) const {
  const int k = static_cast<int>(std::sqrt(static_cast<double>(ord)));   // the reference's own int sqrt(double)
  const int base = k * k;
  if (static_cast<int>(ord) == base) return static_cast<uint16_t>(specials_[k] << 8 | specials_[k]);
  const int off = static_cast<int>(ord) - base - 1;
  if (off < k) return static_cast<uint16_t>(specials_[off] << 8 | specials_[k]);
  return static_cast<uint16_t>(specials_[k] << 8 | specials_[off - k]);
}

void Grammar::expandAlphabet(const std::vector<byte>& freed, const std::vector<byte>& specials, std::vector<uint16_t>* next_pairs) {
  bool renamed[256] = {false};          // symbols that are written as two bytes from now on
  uint16_t as_pair[256] = {0};
  size_t taken = 0;                     // new special symbols used so far
  for (size_t i = 0; i < freed.size(); ++i) {
    while (specialSymbolPairsLeft() == 0) {
      // every pair of the old special symbols is in use: a new special symbol, whose own double stands for itself
      const byte z = specials[taken++];
      pair_use_.push_back(std::make_pair(false, z));
      specials_.push_back(z);
      special_[z] = true;
      renamed[z] = true;
      as_pair[z] = static_cast<uint16_t>(z << 8 | z);
    }
    const uint32_t count = numberOfSpecialSymbols();
    const uint32_t ordinal = static_cast<uint32_t>(pair_use_.size());
    uint32_t index = ordinal - ((count - 1) * (count - 1) + 1);
    const byte f = freed[i];
    const bool was_variable = variable_[f];
    variable_[f] = true;
    uint16_t pair;
    if (index < count - 1) pair = static_cast<uint16_t>(specials_[index] << 8 | specials_.back());    // the newest special symbol second
    else { index = index + 1 - count; pair = static_cast<uint16_t>(specials_.back() << 8 | specials_[index]); }   // ... or first
    pair_use_.push_back(std::make_pair(was_variable, f));
    renamed[f] = true;
    as_pair[f] = pair;
    next_pairs->push_back(pair);
  }
  // the rules made before this round may use a renamed symbol as their variable; every rule may hold one on its right
  const size_t older = rules_.size() - new_rules_;
  for (size_t i = 0; i < rules_.size(); ++i) {
    Rule& r = rules_[i];
    if (i < older && !r.large && renamed[r.variable]) {
      --freq_[r.variable];
      r.variable = as_pair[r.variable];
      r.large = true;
      ++freq_[r.variable >> 8]; ++freq_[r.variable & 0xff];
    }
    std::vector<byte> rhs;
    for (size_t j = 0; j < r.rhs.size(); ++j) {
      const byte b = r.rhs[j];
      if (renamed[b]) {
        --freq_[b];
        ++freq_[as_pair[b] >> 8]; ++freq_[as_pair[b] & 0xff];
        rhs.push_back(static_cast<byte>(as_pair[b] >> 8));
        rhs.push_back(static_cast<byte>(as_pair[b] & 0xff));
      } else rhs.push_back(b);
    }
    r.rhs.swap(rhs);
  }
}

void Grammar::freedSymbols(std::vector<std::pair<uint16_t, byte> >* out) const {
  for (uint32_t i = 0; i < pair_use_.size(); ++i)
    if (!pair_use_[i].first) out->push_back(std::make_pair(specialPair(i), pair_use_[i].second));
}

uint32_t Grammar::freedCount() const {                             // Grammar.cpp:352-364: the doubles (i, i) do not count
  uint32_t n = 0;
  size_t next_square = 0, k = 0;
  for (size_t i = 0; i < pair_use_.size(); ++i) {
    if (i == next_square) { ++k; next_square = k * k; }
    else if (!pair_use_[i].first) ++n;
  }
  return n;
}

static void putPacked(uint64_t v, std::vector<byte>* out) {       // utils::packInteger, low group first
  do { byte b = static_cast<byte>(v & 0x7F); v >>= 7; if (v) b |= 0x80; out->push_back(b); } while (v);
}

void Grammar::write(std::vector<byte>* out) const {               // Grammar.cpp:309-320 and the parts it calls
  putPacked(rules_.size(), out);
  if (rules_.empty()) return;
  out->push_back(static_cast<byte>(specials_.size()));
  out->insert(out->end(), specials_.begin(), specials_.end());
  {                                                               // one flag per rule, first rule in the top bit
    byte acc = 0;
    for (size_t i = 0; i < rules_.size(); ++i) {
      if (rules_[i].large) acc = static_cast<byte>(acc | (0x80u >> (i & 7)));
      if ((i & 7) == 7) { out->push_back(acc); acc = 0; }
    }
    if (rules_.size() & 7) out->push_back(acc);
  }
  for (size_t i = 0; i < rules_.size(); ++i) {
    if (rules_[i].large) out->push_back(static_cast<byte>(rules_[i].variable >> 8));
    out->push_back(static_cast<byte>(rules_[i].variable & 0xff));
  }
  {                                                               // the freed symbols in the order of their pairs (from pair #2 on)
    uint32_t left = freedCount();
    out->push_back(static_cast<byte>(left));
    size_t next_square = 4, k = 2;
    for (size_t i = 2; left > 0; ++i) {
      if (i == next_square) { ++k; next_square = k * k; }
      else if (!pair_use_[i].first) { out->push_back(pair_use_[i].second); --left; }
    }
  }
  {                                                               // lengths 2..4 as two bits each, four rules a byte
    byte acc = 0;
    for (size_t i = 0; i < rules_.size(); ++i) {
      acc = static_cast<byte>(acc << 2 | (rules_[i].rhs.size() - 2));
      if ((i & 3) == 3) { out->push_back(acc); acc = 0; }
    }
    if (rules_.size() & 3) out->push_back(static_cast<byte>(acc << (8 - 2 * (rules_.size() & 3))));
  }
  for (size_t i = 0; i < rules_.size(); ++i) out->insert(out->end(), rules_[i].rhs.begin(), rules_[i].rhs.end());
}

size_t Grammar::read(const byte* in, size_t n) {                  // Grammar.cpp:198-307
  size_t pos = 0;
  bool cut = false;
  struct Src { const byte* p; size_t n; size_t* pos; bool* cut;
               unsigned get() { if (*pos < n) return p[(*pos)++]; *cut = true; ++*pos; return 0; } } src = {in, n, &pos, &cut};
  uint64_t count = 0;
  for (int shift = 0;; shift += 7) { const unsigned b = src.get(); count |= static_cast<uint64_t>(b & 0x7F) << shift; if (!(b & 0x80) || cut) break; }
  if (cut) return static_cast<size_t>(-1);
  if (count == 0) return pos;
  if (count > (1u << 20)) return static_cast<size_t>(-1);
  const size_t ns = src.get();
  int number_of[256] = {0};
  std::vector<bool> stands_for_symbol(ns * ns, true);             // pair #i is a freed symbol's (else: a double, or a variable)
  for (size_t i = 0; i < ns; ++i) {
    const byte z = static_cast<byte>(src.get());
    specials_.push_back(z);
    special_[z] = true;
    number_of[z] = static_cast<int>(i);
    stands_for_symbol[i * i] = false;
  }
  std::vector<bool> large(count);
  for (size_t i = 0; i < count; i += 8) {
    const unsigned flags = src.get();
    for (size_t k = 0; k < 8 && i + k < count; ++k) large[i + k] = (flags >> (7 - k)) & 1u;
  }
  uint32_t highest = 0;
  for (size_t i = 0; i < count; ++i) {
    Rule r;
    r.large = large[i];
    uint16_t var = static_cast<uint16_t>(src.get());
    ++freq_[var];
    if (r.large) {
      const byte second = static_cast<byte>(src.get());
      ++freq_[second];
      const uint32_t ord = ordinalOfPair(static_cast<uint32_t>(number_of[var]), static_cast<uint32_t>(number_of[second]));
      if (ord < stands_for_symbol.size()) stands_for_symbol[ord] = false;
      highest = std::max(highest, ord);
      var = static_cast<uint16_t>(var << 8 | second);
    } else variable_[var] = true;
    r.variable = var;
    rules_.push_back(r);
  }
  const size_t freed = src.get();
  size_t ord = 0, next_square = 0, k = 0, got = 0;
  auto one = [&](bool reading) -> bool {                          // pair #ord
    if (ord == next_square) {
      if (k >= specials_.size()) return false;
      pair_use_.push_back(std::make_pair(false, specials_[k]));
      ++k; next_square = k * k;
    } else if (!reading || (ord < stands_for_symbol.size() && !stands_for_symbol[ord])) {
      pair_use_.push_back(std::make_pair(true, static_cast<byte>(0)));
    } else {
      pair_use_.push_back(std::make_pair(false, static_cast<byte>(src.get())));
      ++got;
    }
    ++ord;
    return true;
  };
  while (got < freed && !cut) if (!one(true)) return static_cast<size_t>(-1);
  if (highest > 0) while (ord <= highest) if (!one(false)) return static_cast<size_t>(-1);
  for (size_t i = 0; i < count; i += 4) {
    const unsigned lens = src.get();
    for (size_t j = 0; j < 4 && i + j < count; ++j) rules_[i + j].rhs.resize(2 + ((lens >> (6 - 2 * j)) & 3u));
  }
  for (size_t i = 0; i < count; ++i)
    for (size_t j = 0; j < rules_[i].rhs.size(); ++j) { rules_[i].rhs[j] = static_cast<byte>(src.get()); ++freq_[rules_[i].rhs[j]]; }
  return cut ? static_cast<size_t>(-1) : pos;
}

// ---- the choice of pairs ---------------------------------------------------------------------------
namespace {

typedef std::pair<byte, uint64_t> SymFreq;
typedef std::pair<uint64_t, uint16_t> PairFreq;

bool lessFrequent(const SymFreq& a, const SymFreq& b) { return a.second < b.second; }

// FrequencyTable (preprocessors/FrequencyTable.cpp): the symbols by ascending frequency.  The order of equally
// frequent symbols is what std::sort leaves (:67-75, an unstable sort on the frequency alone) -- the reference's
// choice of freed symbols, and so its stream, depends on it; same call, same array, same library here.
// remove() (:112-128) moves a symbol behind the live part by giving it the largest frequency plus one.
struct SymbolOrder {
  SymFreq e[256];
  size_t live;
  explicit SymbolOrder(const uint64_t* freq) : live(256) {
    for (int c = 0; c < 256; ++c) e[c] = std::make_pair(static_cast<byte>(c), freq[c]);
    std::sort(e, e + 256, lessFrequent);
  }
  uint32_t frequency(size_t i) const { return static_cast<uint32_t>(e[i].second); }   // (the reference returns uint32 too)
  byte key(size_t i) const { return e[i].first; }
  void remove(byte c) {
    size_t at = 0;
    while (e[at].first != c) ++at;
    const SymFreq moved = std::make_pair(c, e[live - 1].second + 1);
    // FrequencyTable::increase (:130-144): up past every strictly smaller frequency.  (For the very first removal the
    // reference also looks at the entry behind the table -- its own location bytes, read as a count larger than any
    // frequency -- so the walk ends at the table's end either way.)
    while (at < live && at + 1 < 256 && moved.second > e[at + 1].second) { e[at] = e[at + 1]; ++at; }
    e[at] = moved;
    --live;
  }
};

struct Choice { std::vector<PairFreq> pairs; uint32_t variables, specials, for_free; int64_t utility; };

// PairReplacer::findReplaceables (PairReplacer.cpp:130-279): pairs by descending count from `start`, each paid for
// with the next rarest symbol; first as long as unused special pairs (and symbols that do not occur) last, then in
// batches behind every new special symbol (two for the first), a batch kept only when it gains more than 1000.
Choice greedy(size_t start, const std::vector<PairFreq>& pairs, const SymbolOrder& syms, const Grammar& g, size_t max_repl) {
  Choice c;
  c.variables = c.specials = c.for_free = 0;
  c.utility = 0;
  bool is_first[256] = {false}, is_second[256] = {false};
  is_second[0] = true;                                            // sic: `bool usedSnd[256] = {true}` (:136)
  size_t p = start, s = 0;
  auto blocked = [&](byte a, byte b) { return is_first[b] || is_second[a] || g.isSpecial(b) || g.isSpecial(a); };
  uint32_t absent = 0;
  while (absent < syms.live && syms.frequency(absent) == 0) ++absent;
  uint32_t budget = g.specialSymbolPairsLeft() + absent;         // replacements that need no new special symbol
  bool more = p < pairs.size() && s < syms.live && c.pairs.size() < max_repl;
  while (more) {
    if (p >= pairs.size()) break;
    const byte a = static_cast<byte>(pairs[p].second >> 8), b = static_cast<byte>(pairs[p].second & 0xff);
    if (blocked(a, b)) { ++p; continue; }
    const uint32_t cost = syms.frequency(s);
    if (cost + 1003 >= pairs[p].first || budget == 0) break;
    --budget;
    if (cost == 0) ++c.for_free;
    c.utility += static_cast<int64_t>(pairs[p].first) - cost;
    c.pairs.push_back(pairs[p]);
    ++c.variables;
    ++p; ++s;
    is_first[a] = true; is_second[b] = true;
    more = p < pairs.size() && s < syms.live;
  }
  more = p < pairs.size() && s < syms.live && budget == 0;
  uint32_t special_count = g.numberOfSpecialSymbols(), spare = 0;
  while (more) {
    int64_t gain = -static_cast<int64_t>(syms.frequency(s));     // the new special symbol's own occurrences double
    ++s;
    uint32_t limit = 2 * special_count + spare, fresh = 1;
    if (special_count == 0) {
      limit = 2; fresh = 2;
      if (s >= syms.live) break;
      gain -= syms.frequency(s);
      ++s;
    }
    uint32_t batch = 0;
    more = s < syms.live;
    while (batch < limit && more) {
      if (p >= pairs.size()) { more = false; break; }
      const byte a = static_cast<byte>(pairs[p].second >> 8), b = static_cast<byte>(pairs[p].second & 0xff);
      if (blocked(a, b)) { ++p; continue; }
      const uint32_t cost = syms.frequency(s);
      if (cost + 1003 >= pairs[p].first) { more = false; break; }
      gain += static_cast<int64_t>(pairs[p].first) - cost;
      ++batch;
      c.pairs.push_back(pairs[p]);
      ++p; ++s;
      is_first[a] = true; is_second[b] = true;
      more = p < pairs.size() && s < syms.live;
    }
    if (gain > 1000 && batch > 0) {
      c.utility += gain;
      c.variables += batch;
      c.specials += fresh;
      special_count += fresh;
      spare = 0;
    } else {
      more = false;
      c.pairs.resize(c.pairs.size() - batch);
    }
  }
  return c;
}

}  // namespace

void decideReplacements(Grammar* grammar, const uint64_t* byte_freq, const uint64_t* pair_freq, Replacements* out) {
  uint64_t freq[256];
  for (int c = 0; c < 256; ++c) freq[c] = byte_freq[c] + grammar->frequencies()[c];     // PairReplacer.cpp:79-86
  SymbolOrder syms(freq);
  for (int c = 0; c < 256; ++c) if (grammar->isSpecial(static_cast<byte>(c))) syms.remove(static_cast<byte>(c));
  std::vector<PairFreq> pairs(1u << 16);
  for (uint32_t v = 0; v < (1u << 16); ++v) pairs[v] = std::make_pair(pair_freq[v], static_cast<uint16_t>(v));
  std::sort(pairs.rbegin(), pairs.rend());                        // by count, then by pair value, descending (:103)
  Choice best;
  best.variables = best.specials = best.for_free = 0;
  best.utility = 0;
  for (size_t start = 0; start < 5; ++start) {                    // s_greedyStarts
    Choice c = greedy(start, pairs, syms, *grammar, 254);
    if (c.utility > best.utility) best = c;
  }
  const uint32_t from_absent = std::min(best.for_free, best.variables);
  const uint32_t freed_count = best.for_free > best.variables ? 0 : best.variables - best.for_free;
  // the rarest symbols, in order: those that do not occur (used LAST as variables, :440-443), the new special
  // symbols, the symbols to free
  std::vector<byte> absent, specials, freed, variables;
  size_t j = 0;
  for (uint32_t i = 0; i < from_absent; ++i) absent.push_back(syms.key(j++));
  for (uint32_t i = 0; i < best.specials; ++i) specials.push_back(syms.key(j++));
  for (uint32_t i = 0; i < freed_count; ++i) { freed.push_back(syms.key(j)); variables.push_back(syms.key(j)); ++j; }
  variables.insert(variables.end(), absent.begin(), absent.end());
  out->count = static_cast<uint32_t>(best.pairs.size());
  out->common = syms.key(syms.live - 1);
  out->self_pairs = false;
  const uint16_t nothing = static_cast<uint16_t>(out->common << 8 | out->common);
  out->table.assign(1u << 16, nothing);
  if (out->count == 0) return;
  // constructReplacementTable, PairReplacer.cpp:281-328
  grammar->beginUpdatingRules();
  for (uint32_t i = 0; i < out->count; ++i) {
    const uint16_t pair = best.pairs[i].second;
    out->table[pair] = static_cast<uint16_t>(variables[i] << 8 | out->common);
    grammar->addRule(variables[i], static_cast<byte>(pair >> 8), static_cast<byte>(pair & 0xff));
    if ((pair >> 8) == (pair & 0xff)) out->self_pairs = true;
  }
  std::vector<uint16_t> next_pairs;
  grammar->expandAlphabet(freed, specials, &next_pairs);
  auto escape = [&](byte c, uint16_t as) {                        // wherever c starts a pair that is not replaced
    for (uint32_t second = 0; second < 256; ++second) {
      uint16_t& t = out->table[static_cast<uint32_t>(c) << 8 | second];
      if ((t >> 8) == out->common) t = as;
    }
  };
  for (size_t i = 0; i < freed.size(); ++i) escape(freed[i], next_pairs[i]);
  for (size_t i = 0; i < specials.size(); ++i) escape(specials[i], static_cast<uint16_t>(specials[i] << 8 | specials[i]));
  grammar->endUpdatingRules(variables);
}

// ---- Postprocessor ---------------------------------------------------------------------------------
bool postprocess(const Grammar& grammar, const byte* data, size_t n, std::vector<byte>* out, size_t cap) {
  if (grammar.numberOfRules() == 0) {                             // Postprocessor.cpp:114-117
    if (n > cap) return false;
    out->insert(out->end(), data, data + n);
    return true;
  }
  // what every byte, and every pair that starts with a special symbol, stands for (Postprocessor.cpp:62-94)
  std::vector<std::vector<byte> > plain(256), paired(1u << 16);
  for (int c = 0; c < 256; ++c) plain[c].push_back(static_cast<byte>(c));
  {
    std::vector<std::pair<uint16_t, byte> > freed;
    grammar.freedSymbols(&freed);
    for (size_t i = 0; i < freed.size(); ++i) paired[freed[i].first].push_back(freed[i].second);
  }
  auto expand = [&](const byte* src, size_t len, std::vector<byte>* dst, size_t limit) -> bool {
    for (size_t i = 0; i < len; ++i) {
      const std::vector<byte>* e = &plain[src[i]];
      if (grammar.isSpecial(src[i]) && i + 1 < len) { e = &paired[static_cast<uint32_t>(src[i]) << 8 | src[i + 1]]; ++i; }
      if (dst->size() + e->size() > limit) return false;
      dst->insert(dst->end(), e->begin(), e->end());
    }
    return true;
  };
  // The grammar comes out of the stream: a damaged (or hostile) one can chain twenty rules whose right sides name
  // the rule before four times each -- 4^20 bytes.  No rule of a sound grammar stands for more than the block it came
  // from, and all of them together for little more than twice that (a rule is used where it saves bytes).
  const std::vector<Rule>& rules = grammar.rules();
  size_t expanded = 0;
  const size_t budget = 2 * cap + (static_cast<size_t>(1) << 20);
  for (size_t i = 0; i < rules.size(); ++i) {                     // in order: a right side only uses earlier variables
    std::vector<byte> full;
    if (!expand(&rules[i].rhs[0], rules[i].rhs.size(), &full, cap)) return false;
    expanded += full.size();
    if (expanded > budget) return false;
    if (rules[i].large) paired[rules[i].variable].swap(full); else plain[rules[i].variable & 0xff].swap(full);
  }
  return expand(data, n, out, out->size() + cap);
}

// ---- host twins of the GPU passes ------------------------------------------------------------------
// The pair ending at position i counts, except at odd i when it repeats the pair before it (three equal bytes
// in a row; before the text stands a virtual zero byte): PairReplacer.hpp:47-67 unrolled over
// PairReplacer.cpp:53-63.
void pairStatisticsOnHost(const byte* data, size_t n, uint64_t* byte_freq, uint64_t* pair_freq) {
  std::memset(byte_freq, 0, 256 * sizeof(uint64_t));
  std::memset(pair_freq, 0, 65536 * sizeof(uint64_t));
  for (size_t i = 0; i < n; ++i) {
    ++byte_freq[data[i]];
    if (i == 0) continue;
    const byte before = i >= 2 ? data[i - 2] : 0;
    if ((i & 1) && before == data[i - 1] && data[i - 1] == data[i]) continue;
    ++pair_freq[static_cast<uint32_t>(data[i - 1]) << 8 | data[i]];
  }
}

// What position j of the text turns into (PairReplacer.cpp:369-400, position by position): nothing when the pair
// before it was replaced; the variable when the pair it starts is; its escape (two bytes) or itself otherwise.
// Replaced pairs cannot overlap, except along a run of one byte whose double is replaced -- there the pairs are
// taken from the run's start.
size_t writeReplacedOnHost(const Replacements& r, const byte* src, size_t n, byte* dst) {
  const uint16_t nothing = static_cast<uint16_t>(r.common << 8 | r.common);
  size_t out = 0, run_start = 0;
  auto rule_at = [&](size_t j) -> bool {                          // the pair starting at j is one of the replaced pairs
    if (j + 1 >= n) return false;
    const uint16_t t = r.table[static_cast<uint32_t>(src[j]) << 8 | src[j + 1]];
    return t != nothing && (t & 0xff) == r.common;
  };
  bool prev_fired = false;
  for (size_t j = 0; j < n; ++j) {
    if (j > 0 && src[j] != src[j - 1]) run_start = j;
    bool fires = rule_at(j);
    if (fires && src[j] == src[j + 1]) fires = ((j - run_start) & 1) == 0;
    if (prev_fired) { prev_fired = false; continue; }             // consumed as the second byte of a replaced pair
    const uint16_t t = r.table[static_cast<uint32_t>(src[j]) << 8 | (j + 1 < n ? src[j + 1] : 0)];
    if (fires) { dst[out++] = static_cast<byte>(t >> 8); prev_fired = true; }
    else if (j + 1 < n ? t == nothing : (t & 0xff) == r.common) dst[out++] = src[j];
    else { dst[out++] = static_cast<byte>(t >> 8); dst[out++] = static_cast<byte>(t & 0xff); }
  }
  return out;
}

}  // namespace prepr
}  // namespace bwtc
