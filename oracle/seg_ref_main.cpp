// seg_ref_main.cpp -- TEST INFRASTRUCTURE (oracle/): a driver around the GENUINE reference SEG (src/lib/blast/blast_seg.cpp, compiled in
// place into oracle/_ref/obj by the recipe in oracle/Makefile) that mints the golden of tests/test_seg.py. Reads protein FASTA from
// stdin, prints one line per sequence: the id and the intervals (0-based, inclusive) that Masking::operator() replaces with the mask
// letter under --masking seg (src/masking/masking.cpp:172-192), in the order the reference visits them.
// usage: oracle/_ref/seg_ref < proteins.faa > golden.tsv
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>
#include "lib/blast/blast_seg.h"

static void run(const std::string& id, std::vector<unsigned char>& seq)
{
	if (id.empty() && seq.empty()) return;
	SegParameters* sp = SegParametersNewAa();
	BlastSeqLoc* locs = nullptr;
	SeqBufferSeg(seq.data(), (uint32_t)seq.size(), 0u, sp, &locs);
	std::printf("%s", id.c_str());
	for (BlastSeqLoc* l = locs; l; l = l->next) std::printf("\t%d-%d", l->ssr->left, l->ssr->right);
	std::printf("\n");
	SegParametersFree(sp);
}

int main()
{
	static const char* AA = "ARNDCQEGHILKMFPSTWYVBJZX*_";       // amino_acid_traits, src/basic/value.cpp:25
	std::string line, id;
	std::vector<unsigned char> seq;
	bool have = false;
	while (std::getline(std::cin, line)) {
		if (!line.empty() && line.back() == '\r') line.pop_back();
		if (line.empty()) continue;
		if (line[0] == '>') {
			if (have) run(id, seq);
			id = line.substr(1, line.find_first_of(" \t") == std::string::npos ? std::string::npos : line.find_first_of(" \t") - 1);
			seq.clear();
			have = true;
			continue;
		}
		for (char c : line) {
			const char* p = std::strchr(AA, std::toupper((unsigned char)c));
			seq.push_back(p && *p ? (unsigned char)(p - AA) : (std::strchr("UO-uo", c) ? 23 : 23));
		}
	}
	if (have) run(id, seq);
	return 0;
}
