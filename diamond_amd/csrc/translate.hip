// translate.hip -- six-frame translation of DNA reads into the query block of a blastx search (/root/reference/src/util/sequence/translate.h,
// basic/basic.cpp:86-113). Split out of extend_host.hip in round 6.
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>
#include "ctx.h"

using namespace dmnd;

// ---- six-frame translation (blastx query loading) ---------------------------------------------------------------------
namespace {

// NCBI genetic codes (the published translation tables, base order TCAG) that Translator::codes holds (basic/basic.cpp:86-113)
struct GeneticCode { int id; const char* aa; };
const GeneticCode GENETIC_CODES[] = {
	{ 1,  "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG" },     // standard
	{ 2,  "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSS**VVVVAAAADDEEGGGG" },     // vertebrate mitochondrial
	{ 3,  "FFLLSSSSYY**CCWWTTTTPPPPHHQQRRRRIIMMTTTTNNKKSSRRVVVVAAAADDEEGGGG" },     // yeast mitochondrial
	{ 4,  "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG" },     // mold / protozoan mitochondrial, mycoplasma
	{ 5,  "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSSSSVVVVAAAADDEEGGGG" },     // invertebrate mitochondrial
	{ 6,  "FFLLSSSSYYQQCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG" },     // ciliate nuclear
	{ 9,  "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNNKSSSSVVVVAAAADDEEGGGG" },     // echinoderm mitochondrial
	{ 10, "FFLLSSSSYY**CCCWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG" },     // euplotid nuclear
	{ 11, "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG" },     // bacterial, archaeal, plant plastid
	{ 12, "FFLLSSSSYY**CC*WLLLSPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG" },     // alternative yeast nuclear
	{ 13, "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSSGGVVVVAAAADDEEGGGG" },     // ascidian mitochondrial
	{ 14, "FFLLSSSSYYY*CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNNKSSSSVVVVAAAADDEEGGGG" },     // alternative flatworm mitochondrial
	{ 16, "FFLLSSSSYY*LCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG" },     // chlorophycean mitochondrial
	{ 21, "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNNKSSSSVVVVAAAADDEEGGGG" },     // trematode mitochondrial
	{ 22, "FFLLSS*SYY*LCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG" },     // scenedesmus obliquus mitochondrial
	{ 23, "FF*LSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG" },     // thraustochytrium mitochondrial
	{ 24, "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSSKVVVVAAAADDEEGGGG" },     // rhabdopleuridae mitochondrial
	{ 25, "FFLLSSSSYY**CCGWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG" },     // candidate division SR1, gracilibacteria
	{ 26, "FFLLSSSSYY**CC*WLLLAPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG" },     // pachysolen tannophilus nuclear
};

const char* genetic_code(int id)
{
	for (const GeneticCode& g : GENETIC_CODES) if (g.id == id) return g.aa;
	return nullptr;
}

struct CodonTable {
	int8_t fwd[5][5][5], rev[5][5][5];
	explicit CodonTable(const char* code)
	{
		// Translator::init (basic/basic.cpp:116-139): base order TCAG; DNA letters A C G T N = 0..4
		static const char* aa = "ARNDCQEGHILKMFPSTWYVBJZX*_";
		static const int idx[4] = { 2, 1, 3, 0 }, comp[5] = { 3, 2, 1, 0, 4 };
		auto letter = [&](char ch) { return (int8_t)(std::strchr(aa, ch) - aa); };
		for (int i = 0; i < 5; ++i)
			for (int j = 0; j < 5; ++j)
				for (int k = 0; k < 5; ++k) {
					if (i == 4 || j == 4 || k == 4) { fwd[i][j][k] = rev[i][j][k] = 23; continue; }
					fwd[i][j][k] = letter(code[idx[i] * 16 + idx[j] * 4 + idx[k]]);
					rev[i][j][k] = letter(code[idx[comp[i]] * 16 + idx[comp[j]] * 4 + idx[comp[k]]]);
				}
		for (int i = 0; i < 4; ++i)            // an N in the wobble position that cannot change the amino acid
			for (int j = 0; j < 4; ++j) {
				bool f = true, r = true;
				for (int k = 1; k < 4; ++k) { f &= fwd[i][j][k] == fwd[i][j][0]; r &= rev[i][j][k] == rev[i][j][0]; }
				if (f) fwd[i][j][4] = fwd[i][j][0];
				if (r) rev[i][j][4] = rev[i][j][0];
			}
	}
};

// Util::Seq::find_orfs (util/sequence/sequence.cpp:180-197): stretches between stop codons shorter than min_len -> X
void mask_short_orfs(int8_t* s, int n, int min_len)
{
	int begin = 0;
	for (int i = 0; i <= n; ++i)
		if (i == n || s[i] == 24) {
			if (i - begin < min_len) for (int x = begin; x < i; ++x) s[x] = 23;
			begin = i + 1;
		}
}

}

extern "C" int dmnd_translate_opts(const int8_t* dna, int32_t len, int gencode, int strands, int min_orf, int8_t* out[6], int32_t lens[6])
{
	if (!dna || !out || !lens || len < 0) return fail(DMND_E_ARG, "dmnd_translate: bad argument");
	if (strands < 1 || strands > 3) return fail(DMND_E_ARG, "dmnd_translate: strands must be 1 (plus), 2 (minus) or 3 (both)");
	const char* code = genetic_code(gencode);
	if (!code) return fail(DMND_E_ARG, "Invalid genetic code id.");
	static const CodonTable STANDARD(genetic_code(1));
	CodonTable other_storage = STANDARD;
	if (gencode != 1) other_storage = CodonTable(code);
	const CodonTable& T = gencode == 1 ? STANDARD : other_storage;
	for (int f = 0; f < 6; ++f) lens[f] = 0;
	if (len < 3) return DMND_OK;
	for (int32_t i = 0; i < len; ++i)
		if (dna[i] < 0 || dna[i] > 4) return fail(DMND_E_ARG, "dmnd_translate: DNA letters must be 0-4 (ACGTN)");
	for (int f = 0; f < 3; ++f) {
		const int n = (len - f) / 3;
		lens[f] = lens[f + 3] = n;
		for (int i = 0; i < n; ++i) {
			const int p = 3 * i + f;                                    // Translator::getAminoAcid
			out[f][i] = T.fwd[dna[p]][dna[p + 1]][dna[p + 2]];
			const int r = len - 3 - f - 3 * i;                           // Translator::getAminoAcidReverse(dna, r): letters r+2, r+1, r
			out[f + 3][i] = T.rev[dna[r + 2]][dna[r + 1]][dna[r]];
		}
	}
	// config.min_orf_len(frame 0 length): --min-orf, or by read length when it is 0 (basic/config.h:413-424)
	const int l0 = lens[0], min_len = min_orf > 0 ? min_orf : l0 < 30 ? 1 : l0 < 100 ? 20 : 40;
	for (int f = 0; f < 6; ++f) {
		// frames of a strand that is not searched are all mask letters (frame_mask, data/sequence_file.cpp:286-294; Block::push_back, block.cpp:92-99)
		if (strands & (f < 3 ? 1 : 2)) mask_short_orfs(out[f], lens[f], min_len);
		else std::fill(out[f], out[f] + lens[f], (int8_t)23);
	}
	return DMND_OK;
}

extern "C" int dmnd_translate(const int8_t* dna, int32_t len, int8_t* out[6], int32_t lens[6])
{
	return dmnd_translate_opts(dna, len, 1, 3, 0, out, lens);
}

