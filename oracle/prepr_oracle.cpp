/*
 * oracle/prepr_oracle.cpp -- CPU restatement of the reference's pair-replacing pre-stage
 * (`--prepr p...`): preprocessors/PairReplacer.cpp, Grammar.cpp, FrequencyTable.cpp,
 * Precompressor.cpp, Postprocessor.cpp and the block sizing of Compressor.cpp:65-118 when the
 * stage is on.  Every function cites the reference lines it follows (paths under /root/reference).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under bwtc_amd/ links, imports or executes this.
 *
 * C++ rather than C for one reason: FrequencyTable::initialize orders the symbols with
 *   std::sort(m_frequencies, m_frequencies + 256, comparePairSecondAsc)     (FrequencyTable.cpp:67-75)
 * -- an UNSTABLE sort on the frequency alone, so which of several equally rare symbols is freed
 * first is whatever the standard library's introsort does with that array.  The reference's
 * stream with --prepr is therefore only defined up to the libstdc++ it was built with; this file
 * calls the same std::sort on the same array with the same comparator, i.e. gives what the
 * reference gives when built with this toolchain.
 *
 * Pinned by the known answers of the reference's own test/PairReplacerTest.cpp and
 * test/GrammarTest.cpp (tests/golden/prepr_known_answers.json): the reference cannot be built
 * here (globaldefs.hpp includes a Boost header), so no reference-produced stream exists.
 */
#include "bwtc_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

namespace {

typedef uint8_t byte;
typedef uint16_t uint16;
typedef uint32_t uint32;
typedef int64_t int64;

/* ---- FrequencyTable (preprocessors/FrequencyTable.cpp) --------------------------------------- */
struct FreqTable {
  std::pair<byte, size_t> f[256];
  byte loc[256];
  size_t last;

  static bool bySecondAsc(const std::pair<byte, size_t>& a, const std::pair<byte, size_t>& b) { return a.second < b.second; }   /* :36-39 */

  explicit FreqTable(const size_t* freqs) : last(256) {                      /* :63-75 */
    for (int i = 0; i < 256; ++i) f[i] = std::make_pair(static_cast<byte>(i), freqs[i]);
    std::sort(f, f + 256, bySecondAsc);
    for (int i = 0; i < 256; ++i) loc[f[i].first] = static_cast<byte>(i);    /* initLocations :146-150 */
  }
  uint32 getFrequency(size_t i) const { return static_cast<uint32>(f[i].second); }   /* :77-80 */
  byte getKey(size_t i) const { return f[i].first; }                         /* :86-89 */
  size_t size() const { return last; }

  void increase(byte key, size_t value) {                                    /* :130-144 */
    size_t idx = loc[key];
    value += f[idx].second;
    const std::pair<byte, size_t> p = std::make_pair(key, value);
    /* the reference reads f[idx + 1] for idx + 1 == 256 on its first remove(): the bytes behind the
     * array are its location table, read as a count they exceed any frequency, so the walk stops there */
    while (idx < last && idx + 1 < 256 && value > f[idx + 1].second) {
      --loc[f[idx + 1].first];
      f[idx] = f[idx + 1];
      ++idx;
    }
    f[idx] = p;
    loc[key] = static_cast<byte>(idx);
  }
  void remove(byte key) {                                                    /* :112-128 */
    const size_t idx = loc[key];
    increase(key, f[last - 1].second - f[idx].second + 1);
    --last;
  }
};

/* ---- Grammar (preprocessors/Grammar.hpp, Grammar.cpp) ---------------------------------------- */
struct PrRule { uint32 begin, end; uint16 variable; bool large; };

struct Grammar {
  size_t frequencies[256];
  bool isSpecialSymbol[256];
  bool isVariable[256];
  std::vector<byte> specialSymbols;
  std::vector<std::pair<bool, byte> > specialPairReplacements;
  std::vector<PrRule> rules;
  std::vector<byte> rightHandSides;
  uint32 specialSymbolsAsVariables;
  uint16 newRules;

  Grammar() : specialSymbolsAsVariables(0), newRules(0) {                    /* Grammar.cpp:36-40 */
    std::fill(isSpecialSymbol, isSpecialSymbol + 256, false);
    std::fill(frequencies, frequencies + 256, 0);
    std::fill(isVariable, isVariable + 256, false);
  }
  bool isSpecial(byte s) const { return isSpecialSymbol[s]; }
  uint32 specialSymbolPairsLeft() const {                                    /* Grammar.hpp:117-120 */
    const uint32 s = static_cast<uint32>(specialSymbols.size());
    return s * s - static_cast<uint32>(specialPairReplacements.size());
  }
  void addRule(byte variable, byte first, byte second) {                     /* Grammar.cpp:42-51 */
    const uint32 s = static_cast<uint32>(rightHandSides.size());
    PrRule r = {s, s + 2, variable, false};
    rules.push_back(r);
    rightHandSides.push_back(first);
    rightHandSides.push_back(second);
    ++frequencies[first];
    ++frequencies[second];
    ++frequencies[variable];
    ++newRules;
  }

  void expandAlphabet(const std::vector<byte>& freedSymbols, const std::vector<byte>& newSpecials,
                      std::vector<uint16>& nextSpecialPairs) {               /* Grammar.cpp:53-140 */
    bool isNewSpecial[256] = {false};
    uint16 specialReplaces[256] = {0};
    int specialPairsLeft = static_cast<int>(specialSymbolPairsLeft());
    uint32 numOfSpecials = static_cast<uint32>(specialSymbols.size());
    uint32 s = 0;
    for (size_t i = 0; i < freedSymbols.size();) {
      if (specialPairsLeft == 0) {
        specialPairReplacements.push_back(std::make_pair(false, newSpecials[s]));
        specialSymbols.push_back(newSpecials[s]);
        isSpecialSymbol[newSpecials[s]] = true;
        isNewSpecial[newSpecials[s]] = true;
        specialReplaces[newSpecials[s]] = static_cast<uint16>((newSpecials[s] << 8) | newSpecials[s]);
        ++s;
        ++numOfSpecials;
        specialPairsLeft = static_cast<int>(specialSymbolPairsLeft());
      } else {
        const uint32 nextSpecialPair = static_cast<uint32>(specialPairReplacements.size());
        const uint32 offset = (numOfSpecials - 1) * (numOfSpecials - 1) + 1;
        uint32 index = nextSpecialPair - offset;
        const bool toVariable = isVariable[freedSymbols[i]];
        isVariable[freedSymbols[i]] = true;
        if (index < numOfSpecials - 1) {
          specialPairReplacements.push_back(std::make_pair(toVariable, freedSymbols[i]));
          isNewSpecial[freedSymbols[i]] = true;
          const uint16 spPair = static_cast<uint16>((specialSymbols[index] << 8) | specialSymbols.back());
          nextSpecialPairs.push_back(spPair);
          specialReplaces[freedSymbols[i]] = spPair;
        } else {
          index = index + 1 - numOfSpecials;
          specialPairReplacements.push_back(std::make_pair(toVariable, freedSymbols[i]));
          isNewSpecial[freedSymbols[i]] = true;
          const uint16 spPair = static_cast<uint16>((specialSymbols.back() << 8) | specialSymbols[index]);
          specialReplaces[freedSymbols[i]] = spPair;
          nextSpecialPairs.push_back(spPair);
        }
        --specialPairsLeft;
        ++i;
      }
    }
    std::vector<byte> nRightSides;
    const uint32 leftSidesToUpdate = static_cast<uint32>(rules.size()) - newRules;
    for (size_t i = 0; i < rules.size(); ++i) {
      if (i < leftSidesToUpdate) {
        if (!rules[i].large && isNewSpecial[rules[i].variable]) {
          --frequencies[rules[i].variable];
          const uint16 nPair = specialReplaces[rules[i].variable];
          rules[i].variable = nPair;
          rules[i].large = true;
          ++frequencies[nPair >> 8];
          ++frequencies[nPair & 0xff];
          ++specialSymbolsAsVariables;
        }
      }
      const uint32 posInRightSides = static_cast<uint32>(nRightSides.size());
      for (size_t j = rules[i].begin; j < rules[i].end; ++j) {
        if (isNewSpecial[rightHandSides[j]]) {
          --frequencies[rightHandSides[j]];
          const uint16 nPair = specialReplaces[rightHandSides[j]];
          ++frequencies[nPair >> 8];
          ++frequencies[nPair & 0xff];
          nRightSides.push_back(static_cast<byte>(nPair >> 8));
          nRightSides.push_back(static_cast<byte>(nPair & 0xff));
        } else {
          nRightSides.push_back(rightHandSides[j]);
        }
      }
      rules[i].begin = posInRightSides;
      rules[i].end = static_cast<uint32>(nRightSides.size());
    }
    std::swap(nRightSides, rightHandSides);
  }

  uint32 numberOfSpecialPair(uint32 first, uint32 second) const {            /* Grammar.cpp:160-169 */
    if (first == second) return first * second;
    if (first > second) return first * (first + 1) + second + 1;
    return second * second + first + 1;
  }
  uint16 specialPair(uint32 ord) const {                                     /* Grammar.cpp:171-183 */
    const int sqr = static_cast<int>(std::sqrt(static_cast<double>(ord)));
    const int base = sqr * sqr;
    if (static_cast<int>(ord) == base) return static_cast<uint16>((specialSymbols[sqr] << 8) | specialSymbols[sqr]);
    const int offset = static_cast<int>(ord) - base - 1;
    if (offset < sqr) return static_cast<uint16>((specialSymbols[offset] << 8) | specialSymbols[sqr]);
    return static_cast<uint16>((specialSymbols[sqr] << 8) | specialSymbols[offset - sqr]);
  }
  void freedSymbols(std::vector<std::pair<uint16, byte> >& repl) const {     /* Grammar.cpp:185-196 */
    for (uint32 i = 0; i < specialPairReplacements.size(); ++i)
      if (!specialPairReplacements[i].first)
        repl.push_back(std::make_pair(specialPair(i), specialPairReplacements[i].second));
  }
  uint32 numberOfFreedSymbols() const {                                      /* Grammar.cpp:352-364 */
    uint32 n = 0;
    size_t sq = 0, curr = 0;
    for (size_t i = 0; i < specialPairReplacements.size(); ++i) {
      if (sq == i) { ++curr; sq = curr * curr; }
      else if (!specialPairReplacements[i].first) ++n;
    }
    return n;
  }

  /* writeGrammar and its parts, Grammar.cpp:309-462; returns the bytes appended */
  size_t write(std::vector<byte>& out) const {
    const size_t at = out.size();
    {                                                                        /* writeNumberOfRules :446-455 */
      int need;
      uint64_t packed = orc_pack_integer(rules.size(), &need);
      for (int i = 0; i < need; ++i) { out.push_back(static_cast<byte>(packed & 0xff)); packed >>= 8; }
    }
    if (rules.empty()) return out.size() - at;
    out.push_back(static_cast<byte>(specialSymbols.size()));                 /* writeSpecialSymbols :322-330 */
    for (size_t i = 0; i < specialSymbols.size(); ++i) out.push_back(specialSymbols[i]);
    {                                                                        /* writeVariableFlags :424-444 */
      byte buffer = 0, bitsLeft = 8;
      for (size_t i = 0; i < rules.size(); ++i) {
        buffer = static_cast<byte>(buffer | ((rules[i].large ? 1 : 0) << --bitsLeft));
        if (bitsLeft == 0) { out.push_back(buffer); bitsLeft = 8; buffer = 0; }
      }
      if (rules.size() % 8 != 0) out.push_back(buffer);
    }
    for (size_t i = 0; i < rules.size(); ++i) {                              /* writeVariables :337-350 */
      const uint16 var = rules[i].variable;
      if (rules[i].large) out.push_back(static_cast<byte>((var >> 8) & 0xff));
      out.push_back(static_cast<byte>(var & 0xff));
    }
    {                                                                        /* writeFreedSymbols :366-391 */
      uint32 freed = numberOfFreedSymbols();
      out.push_back(static_cast<byte>(freed));
      if (freed > 0) {
        int curr = 2, sq = 4, sqr = 2;
        while (freed > 0) {
          if (sq == curr) { ++sqr; sq = sqr * sqr; }
          else if (!specialPairReplacements[curr].first) { out.push_back(specialPairReplacements[curr].second); --freed; }
          ++curr;
        }
      }
    }
    {                                                                        /* writeLengthsOfRules :393-414 */
      byte buffer = 0;
      uint32 inBuffer = 0;
      for (size_t i = 0; i < rules.size(); ++i) {
        buffer = static_cast<byte>(buffer << 2);
        buffer = static_cast<byte>(buffer | ((rules[i].end - rules[i].begin) - 2));
        if (++inBuffer == 4) { out.push_back(buffer); inBuffer = 0; buffer = 0; }
      }
      if (inBuffer != 0) out.push_back(static_cast<byte>(buffer << (8 - 2 * inBuffer)));
    }
    for (size_t i = 0; i < rightHandSides.size(); ++i) out.push_back(rightHandSides[i]);   /* writeRightSides :416-422 */
    return out.size() - at;
  }

  /* readGrammar, Grammar.cpp:198-307; returns the bytes consumed, (size_t)-1 when the input ends early */
  size_t read(const byte* in, size_t n) {
    size_t pos = 0;
    bool over = false;
    auto rd = [&]() -> unsigned { if (pos < n) return in[pos++]; over = true; ++pos; return 0u; };
    uint32 nrules = 0;
    {                                                                        /* utils::readPackedInteger, Utils.cpp:103-116 */
      int shift = 0;
      for (;;) { const unsigned b = rd(); nrules |= (b & 0x7f) << shift; shift += 7; if (!(b & 0x80) || over) break; }
    }
    if (nrules == 0) return over ? static_cast<size_t>(-1) : pos;
    int specialEnumeration[256] = {0};
    const size_t specials = rd();
    std::vector<bool> original(specials * specials, true);
    for (size_t i = 0; i < specials; ++i) {
      const byte sp = static_cast<byte>(rd());
      specialSymbols.push_back(sp);
      isSpecialSymbol[sp] = true;
      specialEnumeration[sp] = static_cast<int>(i);
      original[i * i] = false;
    }
    uint32 maxSymbol = 0;
    std::vector<bool> isLarge(nrules);
    {                                                                        /* readBit per rule, then flushBuffer */
      unsigned cur = 0; int left = 0;
      for (uint32 i = 0; i < nrules; ++i) {
        if (left == 0) { cur = rd(); left = 8; }
        isLarge[i] = (cur >> --left) & 1u;
      }
    }
    for (uint32 i = 0; i < nrules; ++i) {
      uint16 var = static_cast<uint16>(rd());
      ++frequencies[var];
      if (isLarge[i]) {
        const byte snd = static_cast<byte>(rd());
        ++frequencies[snd];
        const uint32 specEnum = numberOfSpecialPair(static_cast<uint32>(specialEnumeration[var]), static_cast<uint32>(specialEnumeration[snd]));
        if (specEnum < original.size()) original[specEnum] = false;
        if (specEnum > maxSymbol) maxSymbol = specEnum;
        var = static_cast<uint16>((var << 8) | snd);
      } else {
        isVariable[var] = true;
      }
      PrRule r = {0, 0, var, isLarge[i]};
      rules.push_back(r);
    }
    const size_t freed = rd();
    size_t curr = 0, sq = 0, sqr = 0, nread = 0;
    while (nread < freed && !over) {
      if (sq == curr) {
        if (sqr >= specialSymbols.size()) return static_cast<size_t>(-1);
        specialPairReplacements.push_back(std::make_pair(false, specialSymbols[sqr]));
        ++sqr;
        sq = sqr * sqr;
      } else if (curr < original.size() && !original[curr]) {
        specialPairReplacements.push_back(std::make_pair(true, static_cast<byte>(0)));
      } else {
        specialPairReplacements.push_back(std::make_pair(false, static_cast<byte>(rd())));
        ++nread;
      }
      ++curr;
    }
    if (maxSymbol > 0) {
      for (; curr <= maxSymbol; ++curr) {
        if (sq == curr) {
          if (sqr >= specialSymbols.size()) return static_cast<size_t>(-1);
          specialPairReplacements.push_back(std::make_pair(false, specialSymbols[sqr]));
          ++sqr;
          sq = sqr * sqr;
        } else {
          specialPairReplacements.push_back(std::make_pair(true, static_cast<byte>(0)));
        }
      }
    }
    size_t rightSidesLength = 0;
    const size_t add = (nrules % 4 == 0) ? 0 : 1;
    for (size_t i = 0; i < nrules / 4 + add; ++i) {
      const byte lb = static_cast<byte>(rd());
      for (size_t j = 0; j < 4; ++j) {
        const size_t ruleIndex = 4 * i + j;
        if (ruleIndex >= rules.size()) continue;
        const size_t l = 2 + ((lb >> (6 - 2 * j)) & 0x3);
        rules[ruleIndex].begin = static_cast<uint32>(rightSidesLength);
        rules[ruleIndex].end = static_cast<uint32>(rightSidesLength + l);
        rightSidesLength += l;
      }
    }
    rightHandSides.resize(rightSidesLength);
    for (size_t i = 0; i < rightSidesLength; ++i) {
      rightHandSides[i] = static_cast<byte>(rd());
      ++frequencies[rightHandSides[i]];
    }
    return over ? static_cast<size_t>(-1) : pos;
  }
};

/* ---- PairReplacer (preprocessors/PairReplacer.hpp, PairReplacer.cpp) -------------------------- */
typedef std::pair<size_t, uint16> FreqPair;

struct PairReplacer {
  Grammar& g;
  size_t frequencies[256];
  std::vector<size_t> pairFrequencies;     /* 1 << 16 */
  std::vector<uint16> replacements;        /* 1 << 16 */
  uint16 numOfReplacements, numOfFreedSymbols, numOfNewSpecials;
  byte commonByte;

  explicit PairReplacer(Grammar& grammar)
      : g(grammar), pairFrequencies(1 << 16, 0), replacements(1 << 16, 0), numOfReplacements(0),
        numOfFreedSymbols(0), numOfNewSpecials(0), commonByte(0) { std::fill(frequencies, frequencies + 256, 0); }

  /* analyseData(const byte*, size_t) with its helpers, PairReplacer.cpp:53-63,72-86 and
   * PairReplacer.hpp:47-67: every byte is counted; the pair ending at an even position is always
   * counted, the pair ending at an odd position only when it differs from the pair before it */
  void analyseData(const byte* data, size_t length) {
    uint16 prev = data[0];                                                   /* beginAnalysing(first, reset) */
    ++frequencies[data[0]];
    for (int i = 0; i < 256; ++i) frequencies[i] += g.frequencies[i];
    size_t i = 1;
    for (; i < ((length - 1) & 0xfffffffe); ++i) {
      {                                                                      /* analyseData0 */
        ++frequencies[data[i]];
        const uint16 before = prev;
        prev = static_cast<uint16>((prev << 8) | data[i]);
        pairFrequencies[prev] += (before != prev) ? 1 : 0;
      }
      ++i;                                                                   /* analyseData */
      prev = static_cast<uint16>((prev << 8) | data[i]);
      ++pairFrequencies[prev];
      ++frequencies[data[i]];
    }
    if ((length & 0x1) == 0) {
      ++frequencies[data[i]];
      const uint16 before = prev;
      prev = static_cast<uint16>((prev << 8) | data[i]);
      pairFrequencies[prev] += (before != prev) ? 1 : 0;
    }
  }

  /* findReplaceables, PairReplacer.cpp:130-279 */
  int64 findReplaceables(size_t startingPair, const std::vector<FreqPair>& pairs, std::vector<FreqPair>& replPairs,
                         FreqTable& freqs, size_t maxRepl, uint32& variables, uint32& specials, uint32& forFree) const {
    const size_t symbolsToUse = freqs.size();
    bool usedFst[256] = {false}, usedSnd[256] = {true};                      /* sic: usedSnd[0] starts true */
    size_t currentPair = startingPair, currentSymbol = 0;
    int64 utility = 0;
    forFree = 0;
    uint32 vars = 0, specs = 0;
    uint32 freeSymbols = 0;
    while (freeSymbols < freqs.size() && freqs.getFrequency(freeSymbols) == 0) ++freeSymbols;
    uint32 withoutNew = g.specialSymbolPairsLeft() + freeSymbols;
    bool hope = currentPair < pairs.size() && currentSymbol < symbolsToUse && replPairs.size() < maxRepl;
    while (hope) {
      if (currentPair >= pairs.size()) { hope = false; break; }              /* (the reference would read past the list) */
      const byte fst = static_cast<byte>((pairs[currentPair].second >> 8) & 0xFF);
      const byte snd = static_cast<byte>(pairs[currentPair].second & 0xFF);
      if (usedFst[snd] || usedSnd[fst] || g.isSpecial(snd) || g.isSpecial(fst)) { ++currentPair; continue; }
      const uint32 fr = freqs.getFrequency(currentSymbol);
      if (fr + 1003 >= pairs[currentPair].first || withoutNew == 0) { hope = false; break; }
      --withoutNew;
      if (fr == 0) ++forFree;
      utility += static_cast<int64>(pairs[currentPair].first);
      utility -= fr;
      replPairs.push_back(pairs[currentPair]);
      ++vars;
      ++currentPair;
      ++currentSymbol;
      usedFst[fst] = true;
      usedSnd[snd] = true;
      hope = currentPair < pairs.size() && currentSymbol < symbolsToUse;
    }
    hope = currentPair < pairs.size() && currentSymbol < symbolsToUse && withoutNew == 0;
    uint32 tSpecials = static_cast<uint32>(g.specialSymbols.size());
    uint32 unusedSpecials = 0;
    while (hope) {
      int64 utilityAfterNewSpecials = 0;
      const int64 utilityFromFrees = 0, utilityFromFreesBeg = 0;
      uint32 limit = 2 * tSpecials + unusedSpecials;
      uint32 nSpecials = 1;
      utilityAfterNewSpecials -= freqs.getFrequency(currentSymbol);
      currentSymbol++;
      if (tSpecials == 0) {
        limit = 2;
        nSpecials = 2;
        if (currentSymbol >= symbolsToUse) { hope = false; break; }          /* (an alphabet of one symbol) */
        utilityAfterNewSpecials -= freqs.getFrequency(currentSymbol);
        currentSymbol++;
      }
      uint32 pairsAfterSpecial = 0;
      const uint32 freeVars = 0, freeVarsBeg = 0;
      hope = hope && currentSymbol < symbolsToUse;                           /* (the reference reads the table's next entry unchecked) */
      while (pairsAfterSpecial < limit && hope) {
        if (currentPair >= pairs.size()) { hope = false; break; }
        const byte fst = static_cast<byte>((pairs[currentPair].second >> 8) & 0xFF);
        const byte snd = static_cast<byte>(pairs[currentPair].second & 0xFF);
        if (usedFst[snd] || usedSnd[fst] || g.isSpecial(snd) || g.isSpecial(fst)) { ++currentPair; continue; }
        const uint32 fr = freqs.getFrequency(currentSymbol);
        if (fr + 1003 >= pairs[currentPair].first) { hope = false; break; }
        utilityAfterNewSpecials += static_cast<int64>(pairs[currentPair].first);
        utilityAfterNewSpecials -= fr;
        ++pairsAfterSpecial;
        replPairs.push_back(pairs[currentPair]);
        ++currentPair;
        ++currentSymbol;
        usedFst[fst] = true;
        usedSnd[snd] = true;
        hope = currentPair < pairs.size() && currentSymbol < symbolsToUse;
      }
      const int64 totalUt = utilityFromFrees + utilityAfterNewSpecials + utilityFromFreesBeg;
      if (totalUt > 1000 && pairsAfterSpecial > freeVarsBeg) {
        utility += totalUt;
        vars += pairsAfterSpecial + freeVars + freeVarsBeg;
        specs += nSpecials;
        tSpecials += nSpecials;
        unusedSpecials = freeVars + freeVarsBeg;
      } else {
        hope = false;
        for (size_t i = 0; i < pairsAfterSpecial + freeVars + freeVarsBeg; ++i) replPairs.pop_back();
      }
    }
    specials = specs;
    variables = vars;
    return utility;
  }

  /* findReplaceablePairs, PairReplacer.cpp:97-128 */
  void findReplaceablePairs(std::vector<FreqPair>& pairs, std::vector<FreqPair>& replaceablePairs, FreqTable& freqs,
                            size_t maxReplacements, uint32& variables, uint32& specials, uint32& forFree) const {
    std::sort(pairs.rbegin(), pairs.rend());
    int64 bestUtility = 0;
    const FreqTable tFreqs(freqs);
    for (size_t i = 0; i < 5; ++i) {                                         /* s_greedyStarts */
      FreqTable tmpFreqs(tFreqs);
      std::vector<FreqPair> tmpReplacements;
      uint32 tVars = 0, tSpecs = 0, tFree = 0;
      const int64 utility = findReplaceables(i, pairs, tmpReplacements, tmpFreqs, maxReplacements, tVars, tSpecs, tFree);
      if (utility > bestUtility) {
        freqs = tmpFreqs;
        replaceablePairs = tmpReplacements;
        bestUtility = utility;
        variables = tVars;
        specials = tSpecs;
        forFree = tFree;
      }
    }
  }

  /* constructReplacementTable, PairReplacer.cpp:281-328 */
  void constructReplacementTable(const std::vector<FreqPair>& pairs, const std::vector<byte>& freedSymbols,
                                 const std::vector<byte>& newSpecials, const std::vector<byte>& repl) {
    g.newRules = 0;                                                          /* beginUpdatingRules */
    for (size_t i = 0; i < numOfReplacements; ++i) {
      replacements[pairs[i].second] = static_cast<uint16>((repl[i] << 8) | commonByte);
      g.addRule(repl[i], static_cast<byte>((pairs[i].second >> 8) & 0xff), static_cast<byte>(pairs[i].second & 0xff));
    }
    std::vector<uint16> nextSpecialPairs;
    g.expandAlphabet(freedSymbols, newSpecials, nextSpecialPairs);
    for (size_t i = 0; i < freedSymbols.size(); ++i) {
      const uint16 special = nextSpecialPairs[i];
      const uint16 hVal = static_cast<uint16>(freedSymbols[i] << 8);
      for (size_t j = 0; j < 256; ++j)
        if ((replacements[hVal | j] >> 8) == commonByte) replacements[hVal | j] = special;
    }
    for (size_t i = 0; i < newSpecials.size(); ++i) {
      const uint16 special = static_cast<uint16>((newSpecials[i] << 8) | newSpecials[i]);
      const uint16 hVal = static_cast<uint16>(newSpecials[i] << 8);
      for (size_t j = 0; j < 256; ++j)
        if ((replacements[hVal | j] >> 8) == commonByte) replacements[hVal | j] = special;
    }
    for (size_t i = 0; i < repl.size(); ++i) g.isVariable[repl[i]] = true;   /* endUpdatingRules(variables) */
  }

  /* decideReplacements, PairReplacer.cpp:402-484 */
  size_t decideReplacements() {
    FreqTable freqTable(frequencies);
    std::vector<FreqPair> pairs;
    for (size_t i = 0; i < (1u << 16); ++i) pairs.push_back(std::make_pair(pairFrequencies[i], static_cast<uint16>(i)));   /* makePairList */
    std::vector<FreqPair> replaceablePairs;
    for (size_t i = 0; i < 256; ++i) if (g.isSpecial(static_cast<byte>(i))) freqTable.remove(static_cast<byte>(i));
    uint32 nSpecials = 0, nVariables = 0, forFree = 0;
    findReplaceablePairs(pairs, replaceablePairs, freqTable, 254, nVariables, nSpecials, forFree);
    numOfNewSpecials = static_cast<uint16>(nSpecials);
    numOfFreedSymbols = static_cast<uint16>((forFree > nVariables) ? 0 : (nVariables - forFree));
    std::vector<byte> newSpecials, freedSymbols, repl, frees;
    size_t j = 0;
    for (size_t i = 0; i < std::min(forFree, nVariables); ++i) frees.push_back(freqTable.getKey(j++));
    for (size_t i = 0; i < numOfNewSpecials; ++i) newSpecials.push_back(freqTable.getKey(j++));
    for (size_t i = 0; i < numOfFreedSymbols; ++i) { const byte k = freqTable.getKey(j++); freedSymbols.push_back(k); repl.push_back(k); }
    for (size_t i = 0; i < frees.size(); ++i) repl.push_back(frees[i]);
    numOfReplacements = static_cast<uint16>(replaceablePairs.size());
    commonByte = freqTable.getKey(freqTable.size() - 1);
    std::fill(replacements.begin(), replacements.end(), static_cast<uint16>((commonByte << 8) | commonByte));
    if (numOfReplacements > 0) constructReplacementTable(replaceablePairs, freedSymbols, newSpecials, repl);
    return numOfReplacements;
  }

  /* writeReplacedVersion(const byte*, size_t, byte*), PairReplacer.cpp:369-400 (the in-place form,
   * :330-367, writes the same bytes through a ring buffer) */
  size_t writeReplacedVersion(const byte* src, size_t length, byte* dst) const {
    size_t j = 0;
    uint16 pair = src[0];
    size_t i = 1;
    const uint16 noop = static_cast<uint16>((commonByte << 8) | commonByte);
    while (true) {
      pair = static_cast<uint16>((pair << 8) | src[i]);
      uint16 replValue = replacements[pair];
      if (replValue == noop) {
        dst[j++] = src[i - 1];
      } else if ((replValue & 0xff) == commonByte) {
        dst[j++] = static_cast<byte>(replValue >> 8);
        if (i == length - 1) break;
        pair = src[++i];
      } else {
        dst[j++] = static_cast<byte>(replValue >> 8);
        dst[j++] = static_cast<byte>(replValue & 0xff);
      }
      if (i == length - 1) {
        pair = static_cast<uint16>(src[i] << 8);
        if (((replValue = replacements[pair]) & 0xff) != commonByte) {
          dst[j++] = static_cast<byte>(replValue >> 8);
          dst[j++] = static_cast<byte>(replValue & 0xff);
        } else {
          dst[j++] = src[i];
        }
        break;
      }
      ++i;
    }
    return j;
  }
};

/* ---- Postprocessor (preprocessors/Postprocessor.cpp) ------------------------------------------ */
struct Postprocessor {
  std::vector<std::vector<byte> > repl;   /* 1 << 17 */
  bool isSpecial[256];
  bool hasRules;
  explicit Postprocessor(const Grammar& g) : repl(1u << 17), hasRules(!g.rules.empty()) {   /* :62-94 */
    for (size_t i = 0; i < 256; ++i) { repl[i].push_back(static_cast<byte>(i)); isSpecial[i] = g.isSpecialSymbol[i]; }
    const int highBit = 1 << 16;
    std::vector<std::pair<uint16, byte> > freed;
    g.freedSymbols(freed);
    for (size_t i = 0; i < freed.size(); ++i) repl[highBit | freed[i].first].push_back(freed[i].second);
    for (size_t i = 0; i < g.rules.size(); ++i) {
      std::vector<byte> tmp;
      expand(&g.rightHandSides[0] + g.rules[i].begin, g.rules[i].end - g.rules[i].begin, tmp);
      int addr = g.rules[i].variable;
      if (g.rules[i].large) addr |= highBit;
      std::swap(repl[addr], tmp);
    }
  }
  void expand(const byte* src, size_t length, std::vector<byte>& dst) const {   /* :98-110 and :112-133 */
    for (size_t i = 0; i < length; ++i) {
      int key = src[i];
      if (isSpecial[src[i]] && i + 1 < length) { key = (1 << 16) | (src[i] << 8) | src[i + 1]; ++i; }
      dst.insert(dst.end(), repl[key].begin(), repl[key].end());
    }
  }
};

}  // namespace

struct orc_grammar { Grammar g; };

extern "C" {

orc_grammar* orc_grammar_new(void) { return new orc_grammar(); }
void orc_grammar_free(orc_grammar* g) { delete g; }
uint32_t orc_grammar_rules(const orc_grammar* g) { return static_cast<uint32_t>(g->g.rules.size()); }
uint32_t orc_grammar_specials(const orc_grammar* g) { return static_cast<uint32_t>(g->g.specialSymbols.size()); }
int orc_grammar_is_special(const orc_grammar* g, unsigned symbol) { return g->g.isSpecialSymbol[symbol & 255u] ? 1 : 0; }
void orc_grammar_frequencies(const orc_grammar* g, uint64_t f[256]) { for (int i = 0; i < 256; ++i) f[i] = g->g.frequencies[i]; }

size_t orc_grammar_write(const orc_grammar* g, uint8_t* out, size_t cap) {
  std::vector<byte> v;
  g->g.write(v);
  if (v.size() > cap) return static_cast<size_t>(-1);
  if (!v.empty()) std::memcpy(out, &v[0], v.size());
  return v.size();
}
size_t orc_grammar_read(orc_grammar* g, const uint8_t* in, size_t n) { return g->g.read(in, n); }

/* test hook: GrammarTest.cpp's SpecialSymbols case drives addRule / expandAlphabet directly */
void orc_grammar_add_rule(orc_grammar* g, unsigned variable, unsigned first, unsigned second) {
  g->g.addRule(static_cast<byte>(variable), static_cast<byte>(first), static_cast<byte>(second));
}
size_t orc_grammar_expand_alphabet(orc_grammar* g, const uint8_t* freed, size_t n_freed, const uint8_t* specials,
                                   size_t n_specials, uint16_t* next_pairs) {
  std::vector<byte> f(freed, freed + n_freed), s(specials, specials + n_specials);
  std::vector<uint16> np;
  g->g.expandAlphabet(f, s, np);
  for (size_t i = 0; i < np.size(); ++i) next_pairs[i] = np[i];
  return np.size();
}

void orc_pair_statistics(const orc_grammar* g, const uint8_t* data, size_t n, uint64_t freqs[256], uint64_t* pair_freqs) {
  Grammar empty;
  Grammar tmp = g ? g->g : empty;
  PairReplacer pr(tmp);
  pr.analyseData(data, n);
  for (int i = 0; i < 256; ++i) freqs[i] = pr.frequencies[i];
  for (size_t i = 0; i < (1u << 16); ++i) pair_freqs[i] = pr.pairFrequencies[i];
}

/* one PairReplacer over data[0..n): analyseData, decideReplacements, writeReplacedVersion (the PREPROCESS
 * macro, Precompressor.cpp:48-60).  Returns the number of replacements; the new bytes go to out (room for
 * 2 n), their count to *out_n. */
size_t orc_pair_replace_round(orc_grammar* g, const uint8_t* data, size_t n, uint8_t* out, size_t* out_n) {
  PairReplacer pr(g->g);
  pr.analyseData(data, n);
  const size_t rep = pr.decideReplacements();
  *out_n = pr.writeReplacedVersion(data, n, out);
  return rep;
}

/* Precompressor::precompress, Precompressor.cpp:62-121: one round per option letter, stopped by the first
 * round that does not shorten the block.  In place; returns the new length.  Blocks shorter than three
 * bytes are left alone (the reference asserts length > 2 and reads past a shorter block). */
size_t orc_precompress(orc_grammar* g, const char* options, uint8_t* data, size_t n) {
  size_t length = n;
  if (n < 3) return n;
  std::vector<byte> tmp(2 * n + 2);
  for (const char* c = options; c && *c; ++c) {
    const size_t old = length;
    if (*c == 'p') {
      size_t nn = 0;
      orc_pair_replace_round(g, data, length, &tmp[0], &nn);
      std::memcpy(data, &tmp[0], nn);
      length = nn;
    }
    if (length == old) break;
    if (length < 3) break;
  }
  return length;
}

/* Postprocessor::uncompress with the grammar given in its serialised form (one 0x00 byte: no rules).
 * Returns the bytes written or (size_t)-1. */
size_t orc_postprocess_bytes(const uint8_t* gram, size_t gram_n, const uint8_t* data, size_t n, uint8_t* out, size_t cap) {
  Grammar g;
  if (g.read(gram, gram_n) == static_cast<size_t>(-1)) return static_cast<size_t>(-1);
  if (g.rules.empty()) { if (n > cap) return static_cast<size_t>(-1); std::memcpy(out, data, n); return n; }
  Postprocessor post(g);
  std::vector<byte> v;
  v.reserve(2 * n);
  post.expand(data, n, v);
  if (v.size() > cap) return static_cast<size_t>(-1);
  if (!v.empty()) std::memcpy(out, &v[0], v.size());
  return v.size();
}
size_t orc_postprocess(const orc_grammar* g, const uint8_t* data, size_t n, uint8_t* out, size_t cap) {
  std::vector<byte> gram;
  g->g.write(gram);
  return orc_postprocess_bytes(&gram[0], gram.size(), data, n, out, cap);
}
/* how many bytes the serialised grammar at in[0..n) occupies, (size_t)-1 when it is cut short */
size_t orc_grammar_span(const uint8_t* in, size_t n) { Grammar g; return g.read(in, n); }

/* Compressor::compress with the pre-stage on (Compressor.cpp:65-118): precompressor blocks of 0.74 memLimit
 * bytes, each precompressed, then cut into BWT blocks of (memLimit - precompressed size) / 4.5 bytes
 * (:94-97), header = packed original size, packed slices, grammar (PrecompressorBlock.cpp:64-90).
 * coder: 'H' or a wavelet letter. */
size_t orc_compress_prepr(char coder, const char* options, const uint8_t* in, size_t size, uint64_t mem_limit,
                          uint32_t starting_points, uint8_t* out, size_t out_cap) {
  size_t pos = 0, off = 0;
  const bool prepr = options && options[0];
  size_t pbBlockSize = static_cast<size_t>(mem_limit * 0.74);
  size_t bwtBlockSize = std::min(static_cast<size_t>(mem_limit * 0.185), static_cast<size_t>(0x7fffffff - 1));
  if (!prepr) pbBlockSize = bwtBlockSize;
  if (pbBlockSize == 0) return 0;
  void* ws = coder == 'H' ? nullptr : orc_wavelet_stream_new(coder);
  if (coder != 'H' && !ws) return 0;
  out[pos++] = static_cast<uint8_t>(coder);
  while (off < size) {
    const size_t orig = std::min(pbBlockSize, size - off);
    std::vector<byte> blk(in + off, in + off + orig);
    blk.resize(orig + 1);
    orc_grammar g;
    size_t len = orig;
    if (prepr) len = orc_precompress(&g, options, &blk[0], orig);
    if (pbBlockSize != bwtBlockSize) {
      const size_t s = static_cast<size_t>((mem_limit - len) / 4.5);
      bwtBlockSize = std::min(s, static_cast<size_t>(0x7fffffff - 1));
    }
    if (bwtBlockSize == 0) { if (ws) orc_wavelet_stream_free(ws); return 0; }
    const size_t slices = (len + bwtBlockSize - 1) / bwtBlockSize;
    int need;
    uint64_t packed = orc_pack_integer(orig, &need);
    for (int i = 0; i < need; ++i) { out[pos++] = static_cast<uint8_t>(packed & 0xff); packed >>= 8; }
    packed = orc_pack_integer(slices, &need);
    for (int i = 0; i < need; ++i) { out[pos++] = static_cast<uint8_t>(packed & 0xff); packed >>= 8; }
    {
      std::vector<byte> gram;
      g.g.write(gram);
      std::memcpy(out + pos, &gram[0], gram.size());
      pos += gram.size();
    }
    for (size_t s = 0; s < slices; ++s) {
      const size_t b0 = s * bwtBlockSize, bs = std::min(bwtBlockSize, len - b0);
      std::vector<byte> slice(blk.begin() + b0, blk.begin() + b0 + bs);
      slice.push_back(0);
      uint32_t lf[256], n_lf = 0, freqs[256];
      std::memset(freqs, 0, sizeof freqs);
      orc_bwt_block(&slice[0], static_cast<uint32_t>(bs), starting_points, lf, &n_lf, freqs);
      if (coder == 'H') pos += orc_huffman_encode_block(&slice[0], static_cast<uint32_t>(bs), lf, n_lf, freqs, out + pos, out_cap - pos);
      else pos += orc_wavelet_stream_encode_block(ws, &slice[0], static_cast<uint32_t>(bs), lf, n_lf, freqs, out + pos);
    }
    off += orig;
  }
  out[pos++] = 0;
  if (ws) orc_wavelet_stream_free(ws);
  return pos;
}

}  // extern "C"
