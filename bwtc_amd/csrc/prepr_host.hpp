// Host half of the pair-replacing pre-stage (`--prepr p...`): the grammar, the greedy choice of pairs and the
// replacement table, the postprocessor.  Replaces preprocessors/Grammar.{hpp,cpp}, the deciding half of
// preprocessors/PairReplacer.cpp (:97-328, :402-484), FrequencyTable.cpp and Postprocessor.cpp of the
// reference; the two data-parallel halves -- pair statistics (PairReplacer.cpp:53-63, PairReplacer.hpp:47-67)
// and the replaced text (PairReplacer.cpp:369-400) -- run on the GPU (prepr.hip).  HIP-free.
#pragma once
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

namespace bwtc {
namespace prepr {

typedef uint8_t byte;

// One rule of the grammar: variable -> right-hand side.  A "large" variable is a pair of special symbols
// (Grammar.hpp:50-84).
struct Rule {
  uint16_t variable;
  bool large;
  std::vector<byte> rhs;
};

// bwtc::Grammar: what the pre-stage did to a precompressor block, carried from round to round and written into
// the block's header (PrecompressorBlock.cpp:64-90).
class Grammar {
 public:
  Grammar();
  bool isSpecial(byte s) const { return special_[s]; }
  bool isVariable(byte s) const { return variable_[s]; }
  uint32_t numberOfRules() const { return static_cast<uint32_t>(rules_.size()); }
  uint32_t numberOfSpecialSymbols() const { return static_cast<uint32_t>(specials_.size()); }
  uint32_t specialSymbolPairsLeft() const {
    const uint32_t s = numberOfSpecialSymbols();
    return s * s - static_cast<uint32_t>(pair_use_.size());
  }
  const uint64_t* frequencies() const { return freq_; }          // bytes of the rules (both sides), Grammar.hpp:134-136
  const std::vector<Rule>& rules() const { return rules_; }

  // one round's update (PairReplacer::constructReplacementTable's calls): the new rules, then the alphabet
  void beginUpdatingRules() { new_rules_ = 0; }
  void addRule(byte variable, byte first, byte second);
  // freed[i] becomes the special pair next_pairs[i]; new special symbols are taken from `specials` as the
  // pairs of the old ones run out (Grammar.cpp:53-140)
  void expandAlphabet(const std::vector<byte>& freed, const std::vector<byte>& specials, std::vector<uint16_t>* next_pairs);
  void endUpdatingRules(const std::vector<byte>& variables) { for (size_t i = 0; i < variables.size(); ++i) variable_[variables[i]] = true; }

  // (special pair, the original symbol it stands for), Grammar.cpp:185-196
  void freedSymbols(std::vector<std::pair<uint16_t, byte> >* out) const;

  void write(std::vector<byte>* out) const;                      // Grammar::writeGrammar, appended
  // Grammar::readGrammar into an empty grammar: bytes consumed, or (size_t)-1 when the input is cut short
  size_t read(const byte* in, size_t n);

 private:
  uint16_t specialPair(uint32_t ord) const;
  uint32_t ordinalOfPair(uint32_t first, uint32_t second) const;
  uint32_t freedCount() const;
  uint64_t freq_[256];
  bool special_[256], variable_[256];
  std::vector<byte> specials_;                                   // in the order they were made
  std::vector<std::pair<bool, byte> > pair_use_;                 // special pair #i: (used as a grammar variable, else the freed symbol)
  std::vector<Rule> rules_;
  uint32_t new_rules_;
};

// What one PairReplacer decided: the table the text is rewritten with.  table[first << 8 | second]:
//   (common, common)  nothing to do at this pair
//   (v, common)       the pair is replaced by v
//   anything else     `first` is written as these two bytes (a freed symbol or a new special symbol)
struct Replacements {
  std::vector<uint16_t> table;       // 65536 entries
  byte common;
  uint32_t count;                    // pairs replaced
  bool self_pairs;                   // some rule replaces a pair of equal bytes (runs need their parity)
};

// PairReplacer::decideReplacements (PairReplacer.cpp:402-484) over the statistics of the text:
// byte_freq[256] WITHOUT the grammar's own bytes (added here, PairReplacer.cpp:79-86), pair_freq[65536].
// Updates the grammar.
void decideReplacements(Grammar* grammar, const uint64_t* byte_freq, const uint64_t* pair_freq, Replacements* out);

// Postprocessor (Postprocessor.cpp:62-133): expands `data`; false when the output does not fit `cap`.
bool postprocess(const Grammar& grammar, const byte* data, size_t n, std::vector<byte>* out, size_t cap);

// Host twins of the two GPU passes (the CPU suite, and blocks too short for a launch to pay).
void pairStatisticsOnHost(const byte* data, size_t n, uint64_t* byte_freq, uint64_t* pair_freq);
size_t writeReplacedOnHost(const Replacements& r, const byte* src, size_t n, byte* dst);

}  // namespace prepr
}  // namespace bwtc
