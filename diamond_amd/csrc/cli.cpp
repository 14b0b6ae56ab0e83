// cli.cpp -- `diamond-hip`: the reference's command line for the hot path (makedb / blastp) on top of the C ABI in
// include/diamond_hip.h. Host I/O only (FASTA and .dmnd in, BLAST tabular out); every alignment step runs through
// libdiamond_hip.so on the MI355X.
// Mirrors: option names of src/basic/config.cpp:217-345, the FASTA reader's letter mapping (src/basic/value.cpp:25-47,
// amino-acid alphabet "ARNDCQEGHILKMFPSTWYVBJZX*_", U/O/- -> X), the .dmnd layout (src/legacy/dmnd/dmnd.cpp:50-117,224-340:
// ReferenceHeader 40 bytes, ReferenceHeader2 as a size-prefixed record, per sequence 0xFF letters 0xFF id 0, trailer of
// (pos u64, len u32, pad u32) records), SequenceSet layout (src/data/string_set.h:27-60), tabular output
// (src/output/blast_tab_format.cpp, sequence ids cut at the first blank).
// Supported: blastp / blastx (--fast, default sensitivity, --sensitive), tantan masking on the GPU (default) or --masking 0
// motif soft masking (default; table in motifs.bin next to the binary), --algo 0 / 1 / auto; SEG is not part of this build; -e, -k, -p,
// -f 6 [FIELD...] (BLAST tabular with the reference's field names) and -f 0 (BLAST pairwise),
// -b / -c: query and reference blocks cut as load_seqs cuts them, records of a query block merged over the reference blocks as
// join_blocks does (output/join_blocks.cpp) -> same text as the reference run with the same -b.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <future>
#include <iostream>
#include <mutex>
#include <thread>
#include <stdexcept>
#include <string>
#include <vector>
#include <future>
#include <zlib.h>
#include <unistd.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include "../../include/diamond_hip.h"

namespace {

// vector storage that is not zero-filled on resize: a reference block of hundreds of MB is written exactly once, by the loader's
// threads (the fill would be one more single-threaded pass over it, and the first touch of every page)
template<typename T> struct NoInitAlloc : std::allocator<T> {
	template<typename U> struct rebind { using other = NoInitAlloc<U>; };
	NoInitAlloc() = default;
	template<typename U> NoInitAlloc(const NoInitAlloc<U>&) {}
	template<typename U> void construct(U* p) { ::new ((void*)p) U; }
	template<typename U, typename... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};
using Letters = std::vector<int8_t, NoInitAlloc<int8_t>>;

struct SeqBlock {
	Letters data;                    // SequenceSet layout
	std::vector<int64_t> limits;
	std::vector<std::string> ids;
	int64_t letters = 0;
	void begin() { data.assign(256, 31); limits.assign(1, 256); }
	void push(const std::vector<int8_t>& s, const std::string& id)
	{
		data.insert(data.end(), s.begin(), s.end());
		data.push_back(31);
		limits.push_back((int64_t)data.size());
		ids.push_back(id);
		letters += (int64_t)s.size();
	}
	void finish() { data.insert(data.end(), 256, 31); }
};

int8_t letter_of(char c)
{
	static int8_t map[256];
	static bool init = false;
	if (!init) {
		std::memset(map, -1, sizeof map);
		const char* alpha = "ARNDCQEGHILKMFPSTWYVBJZX*_";
		for (int i = 0; alpha[i]; ++i) { map[(unsigned char)alpha[i]] = (int8_t)i; map[(unsigned char)std::tolower(alpha[i])] = (int8_t)i; }
		for (const char* m = "UO-"; *m; ++m) { map[(unsigned char)*m] = 23; map[(unsigned char)std::tolower(*m)] = 23; }
		init = true;
	}
	return map[(unsigned char)c];
}

std::string short_id(const std::string& title)
{
	const size_t e = title.find_first_of(" \t");
	return e == std::string::npos ? title : title.substr(0, e);
}

// Sequence file -> text in memory. Compressed input is detected by the gzip magic as the reference's InputFile does
// (File::Flags::DETECT_COMPRESSION, data/fasta/fasta_file.cpp:68; zlib inflates it, also concatenated members); FASTQ
// (first character '@', guess_format fasta_file.cpp:43-52) is rewritten as FASTA text -- title, letters up to the '+' line, as many
// quality characters as letters skipped (FastqTokenizer::read_record, data/fasta/parser.h:238-270).
std::vector<char> read_sequence_text(const std::string& path)
{
	std::vector<char> file;
	unsigned char magic[2] = { 0, 0 };
	const bool from_stdin = path.empty() || path == "-";       // File::Flags::TREAT_BLANK_AS_STDIN: zlib reads plain and gzip data alike
	if (from_stdin) { magic[0] = 0x1f; magic[1] = 0x8b; }
	else {
		std::ifstream f(path, std::ios::binary | std::ios::ate);
		if (!f) throw std::runtime_error("Error opening file " + path);
		const std::streamoff n = f.tellg();
		f.seekg(0);
		if (n >= 2) { f.read((char*)magic, 2); f.seekg(0); }
		if (!(magic[0] == 0x1f && magic[1] == 0x8b)) {
			file.resize((size_t)n);
			if (n > 0 && !f.read(file.data(), n)) throw std::runtime_error("Error reading file " + path);
		}
	}
	if (magic[0] == 0x1f && magic[1] == 0x8b) {
		gzFile g = from_stdin ? gzdopen(dup(0), "rb") : gzopen(path.c_str(), "rb");
		if (!g) throw std::runtime_error("Error opening file " + (from_stdin ? std::string("(standard input)") : path));
		gzbuffer(g, 1 << 20);
		size_t have = 0;
		for (;;) {
			if (file.size() < have + (1 << 22)) file.resize(std::max<size_t>(file.size() * 2, have + (1 << 22)));
			const int got = gzread(g, file.data() + have, 1 << 22);
			if (got < 0) { gzclose(g); throw std::runtime_error("Error reading compressed file " + path); }
			if (got == 0) break;
			have += (size_t)got;
		}
		gzclose(g);
		file.resize(have);
	}
	if (file.empty()) throw std::runtime_error("Error detecting input file format. Input file seems to be empty.");
	if (file[0] != '>' && file[0] != '@')
		throw std::runtime_error("Error detecting input file format (when treating as text file). First line must begin with '>' (FASTA) or '@' (FASTQ).");
	if (file[0] == '>') return file;
	std::vector<char> fasta;
	fasta.reserve(file.size() / 2 + 16);
	const char* p = file.data();
	const char* const end = p + file.size();
	int64_t lineno = 0;
	auto line = [&](const char*& b, const char*& e) -> bool {            // next line without its terminator; false at the end of the file
		if (p >= end) return false;
		const char* nl = (const char*)std::memchr(p, '\n', (size_t)(end - p));
		b = p; e = nl ? nl : end; p = nl ? nl + 1 : end;
		if (e > b && e[-1] == '\r') --e;
		++lineno;
		return true;
	};
	const char *b, *e;
	while (line(b, e)) {
		if (b == e && p >= end) break;
		const int64_t at = lineno;
		auto bad = [&] { return std::runtime_error("Malformed FASTQ record at line " + std::to_string(at)); };
		if (b == e || *b != '@') throw bad();
		fasta.push_back('>');
		fasta.insert(fasta.end(), b + 1, e);
		fasta.push_back('\n');
		int64_t len = 0, qlen = 0;
		for (;;) {
			if (!line(b, e) || b == e) throw bad();
			if (*b == '+') break;
			len += e - b;
			fasta.insert(fasta.end(), b, e);
		}
		fasta.push_back('\n');
		while (qlen < len && line(b, e)) qlen += e - b;
	}
	return fasta;
}

void read_fasta(const std::string& path, SeqBlock& b)
{
	const std::vector<char> file = read_sequence_text(path);
	b.begin();
	b.data.reserve(256 + file.size() + 256);
	letter_of('A');                                        // builds the letter map
	bool have = false;
	size_t seq_begin = 0;
	std::string id;
	auto flush = [&] {
		if (!have) return;
		if (b.data.size() == seq_begin) throw std::runtime_error("File format error: sequence of length 0");
		b.letters += (int64_t)(b.data.size() - seq_begin);
		b.data.push_back(31);
		b.limits.push_back((int64_t)b.data.size());
		b.ids.push_back(id);
	};
	const char* p = file.data();
	const char* const end = p + file.size();
	while (p < end) {
		const char* nl = (const char*)std::memchr(p, '\n', (size_t)(end - p));
		const char* le = nl ? nl : end;
		const char* next = nl ? nl + 1 : end;
		if (le > p && le[-1] == '\r') --le;
		if (le > p) {
			if (*p == '>') { flush(); id.assign(p + 1, le); have = true; seq_begin = b.data.size(); }
			else {
				if (!have) throw std::runtime_error("FASTA format error: missing '>' in " + path);
				for (const char* c = p; c < le; ++c) {
					if (*c == ' ' || *c == '\t') continue;
					const int8_t l = letter_of(*c);
					if (l < 0) throw std::runtime_error(std::string("Invalid character (") + *c + ") in sequence " + id);
					b.data.push_back(l);
				}
			}
		}
		p = next;
	}
	flush();
	b.finish();
}

// blastx query file: DNA reads -> six translated frames per read, consecutive in the block (Block::push_back,
// data/block/block.cpp:82-100). source_len keeps the read lengths for the DNA coordinates of the output.
void read_dna_fasta_translated(const std::string& path, SeqBlock& b, std::vector<int32_t>& source_len, std::vector<std::string>& read_ids, std::vector<std::vector<int8_t>>& reads,
	int gencode, int strands, int min_orf)
{
	const std::vector<char> file = read_sequence_text(path);
	static int8_t map[256];
	static bool init = false;
	if (!init) {                                                     // nucleotide_traits("ACGTN", 4, "MRWSYKVHDBX"), stats/stats.cpp:42
		std::memset(map, -1, sizeof map);
		const char* alpha = "ACGTN";
		for (int i = 0; alpha[i]; ++i) { map[(unsigned char)alpha[i]] = (int8_t)i; map[(unsigned char)std::tolower(alpha[i])] = (int8_t)i; }
		for (const char* m = "MRWSYKVHDBX"; *m; ++m) { map[(unsigned char)*m] = 4; map[(unsigned char)std::tolower(*m)] = 4; }
		init = true;
	}
	b.begin();
	std::string line, id;
	std::vector<int8_t> seq, frames[6];
	bool have = false;
	auto flush = [&] {
		if (!have) return;
		if (seq.empty()) throw std::runtime_error("File format error: sequence of length 0");
		int8_t* out[6];
		int32_t lens[6];
		for (int k = 0; k < 6; ++k) { frames[k].assign(seq.size() / 3 + 1, 0); out[k] = frames[k].data(); }
		if (dmnd_translate_opts(seq.data(), (int32_t)seq.size(), gencode, strands, min_orf, out, lens) != DMND_OK) throw std::runtime_error(dmnd_last_error());
		for (int k = 0; k < 6; ++k) { frames[k].resize((size_t)lens[k]); b.push(frames[k], id); }
		source_len.push_back((int32_t)seq.size());
		read_ids.push_back(id);
		reads.push_back(seq);
		seq.clear();
	};
	const char* p = file.data();
	const char* const end = p + file.size();
	while (p < end) {
		const char* nl = (const char*)std::memchr(p, '\n', (size_t)(end - p));
		const char* le = nl ? nl : end;
		line.assign(p, le);
		p = nl ? nl + 1 : end;
		if (!line.empty() && line.back() == '\r') line.pop_back();
		if (line.empty()) continue;
		if (line[0] == '>') { flush(); id = line.substr(1); have = true; continue; }
		if (!have) throw std::runtime_error("FASTA format error: missing '>' in " + path);
		for (char c : line) {
			if (c == ' ' || c == '\t') continue;
			const int8_t l = map[(unsigned char)c];
			if (l < 0) throw std::runtime_error(std::string("Invalid character (") + c + ") in sequence " + id);
			seq.push_back(l);
		}
	}
	flush();
	b.finish();
}

const uint64_t DMND_MAGIC = 0x24af8a415ee186dULL;

bool is_dmnd(const std::string& path)
{
	std::ifstream f(path, std::ios::binary);
	uint64_t m = 0;
	f.read((char*)&m, 8);
	return f && m == DMND_MAGIC;
}

SeqBlock slice(const SeqBlock& all, size_t begin, size_t end);

// The reference database as the block loop sees it. A .dmnd file is NOT read into memory as a whole: the header and the position
// array give every sequence's length and file offset, a reference block is the byte range of its records (one read, parsed into the
// SequenceSet layout), and the next block of a GPU's share is read by a helper thread while the current one is searched -- host
// memory holds two blocks per GPU instead of the database (.dmnd streaming). Titles and unmasked sequences of the
// reported targets are read on demand. A FASTA database is parsed once and sliced.
struct Database {
	bool dmnd = false;
	std::string path;
	SeqBlock all;                               // FASTA input
	std::vector<uint64_t> pos;                  // .dmnd: record offsets, n + 1 (the last = position array offset)
	std::vector<uint32_t> len;
	size_t n = 0;
	int64_t letters = 0;
	mutable std::mutex mtx;
	mutable std::vector<std::string> title_cache;
	mutable std::vector<char> have_title;
	const char* map = nullptr;                  // .dmnd: the file, mapped read-only (the page cache is the only copy on the host
	size_t map_size = 0;                        // until a block's letters are written out in the SequenceSet layout)

	Database() = default;
	Database(const Database&) = delete;
	~Database() { if (map) ::munmap(const_cast<char*>(map), map_size); }

	void need(uint64_t at, uint64_t bytes) const { if (at > map_size || bytes > map_size - at) throw std::runtime_error("Truncated DIAMOND database."); }
	void open(const std::string& p)
	{
		path = p;
		dmnd = is_dmnd(p);
		if (!dmnd) { read_fasta(p, all); n = all.ids.size(); letters = all.letters; return; }
		const int fd = ::open(p.c_str(), O_RDONLY);
		if (fd < 0) throw std::runtime_error("Error opening file " + p);
		struct stat st;
		if (::fstat(fd, &st) != 0 || st.st_size < 40) { ::close(fd); throw std::runtime_error("Database file is not a DIAMOND database."); }
		map_size = (size_t)st.st_size;
		void* m = ::mmap(nullptr, map_size, PROT_READ, MAP_PRIVATE, fd, 0);
		::close(fd);
		if (m == MAP_FAILED) { map_size = 0; throw std::runtime_error("Error mapping file " + p); }
		map = static_cast<const char*>(m);
		uint64_t magic, sequences, let, pos_array_offset; uint32_t build, version;
		std::memcpy(&magic, map, 8); std::memcpy(&build, map + 8, 4); std::memcpy(&version, map + 12, 4);
		std::memcpy(&sequences, map + 16, 8); std::memcpy(&let, map + 24, 8); std::memcpy(&pos_array_offset, map + 32, 8);
		(void)build;
		if (magic != DMND_MAGIC) throw std::runtime_error("Database file is not a DIAMOND database.");
		if (version < 2 || version > 3) throw std::runtime_error("Unsupported DIAMOND database version (protein databases of format 2-3 only).");
		n = (size_t)sequences; letters = (int64_t)let;
		need(pos_array_offset, (uint64_t)(n + 1) * 16);
		const char* pa = map + pos_array_offset;
		pos.resize(n + 1); len.resize(n + 1);
		for (size_t i = 0; i <= n; ++i) { std::memcpy(&pos[i], pa + 16 * i, 8); std::memcpy(&len[i], pa + 16 * i + 8, 4); }
		for (size_t i = 0; i < n; ++i)               // every record inside the file, in order: the loader's threads index with these
			if (pos[i + 1] < pos[i] || pos[i + 1] > map_size || (uint64_t)len[i] + 3 > pos[i + 1] - pos[i]) throw std::runtime_error("Truncated DIAMOND database.");
		title_cache.resize(n); have_title.assign(n, 0);
	}
	int64_t length(size_t i) const { return dmnd ? (int64_t)len[i] : all.limits[i + 1] - all.limits[i] - 1; }
	// sequences [begin, end) as a block of their own (SequenceSet layout with its padding). The reference reads the records of a
	// block one after the other on one thread (load_seqs, data/sequence_file.cpp:113-150 over legacy/dmnd/dmnd.cpp:224-340); here
	// the position array gives every sequence's place in the block up front, so `threads` threads each write their share of the
	// block straight from the mapped file -- one pass over the letters, no intermediate copy.
	SeqBlock load(size_t begin, size_t end, int threads = 8) const
	{
		if (!dmnd) return slice(all, begin, end);
		SeqBlock b;
		const size_t count = end - begin;
		b.limits.resize(count + 1);
		b.limits[0] = 256;
		for (size_t k = 0; k < count; ++k) b.limits[k + 1] = b.limits[k] + (int64_t)len[begin + k] + 1;
		const int64_t total = b.limits[count] - 256 - (int64_t)count;
		b.data.resize((size_t)b.limits[count] + 256);
		int8_t* const data = b.data.data();
		// hundreds of MB that are written once, front to back: huge pages cut the page faults of the first touch by 512
		// (a hint; ignored where transparent huge pages are off)
		{
			const uintptr_t a0 = ((uintptr_t)data + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1), a1 = ((uintptr_t)data + b.data.size()) & ~(uintptr_t)((2u << 20) - 1);
			if (a1 > a0) (void)::madvise((void*)a0, a1 - a0, MADV_HUGEPAGE);
		}
		std::memset(data, 31, 256);
		std::memset(data + b.limits[count], 31, 256);
		const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, threads), count / 4096 + 1));
		auto work = [&](int t) {
			// shares of about equal letters: the sequence whose offset passes t / T of the block starts share t
			const int64_t lo_off = 256 + (b.limits[count] - 256) * t / T, hi_off = 256 + (b.limits[count] - 256) * (t + 1) / T;
			size_t k0 = (size_t)(std::lower_bound(b.limits.begin(), b.limits.end() - 1, lo_off) - b.limits.begin());
			size_t k1 = t + 1 == T ? count : (size_t)(std::lower_bound(b.limits.begin(), b.limits.end() - 1, hi_off) - b.limits.begin());
			for (size_t k = k0; k < k1; ++k) {
				const uint32_t l = len[begin + k];
				int8_t* dst = data + b.limits[k];
				const char* src = map + pos[begin + k] + 1;       // record: 0xFF letters 0xFF id 0
				// the reference's makedb stores its SEG soft mask in bit 7 (src/legacy/dmnd/dmnd.cpp:262-265); with masking off
				// the search ignores it (Sequence::operator[] & LETTER_MASK), so it is dropped at load time
				for (uint32_t x = 0; x < l; ++x) dst[x] = (int8_t)(src[x] & 31);
				dst[l] = 31;
			}
		};
		if (T == 1) work(0);
		else {
			std::vector<std::thread> th;
			for (int t = 0; t < T; ++t) th.emplace_back(work, t);
			for (auto& x : th) x.join();
		}
		b.letters = total;
		return b;
	}
	const std::string& title(size_t i) const
	{
		if (!dmnd) return all.ids[i];
		// the formatting threads ask for a title per record, nearly always a different one: a state per title (0 absent, 1 being
		// written, 2 there) instead of one lock for all
		char* state = &have_title[i];
		for (;;) {
			const char s = __atomic_load_n(state, __ATOMIC_ACQUIRE);
			if (s == 2) return title_cache[i];
			char expect = 0;
			if (s == 0 && __atomic_compare_exchange_n(state, &expect, (char)1, false, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED)) {
				const uint64_t at = pos[i] + (uint64_t)len[i] + 2, sz = pos[i + 1] - at;
				title_cache[i].assign(map + at, strnlen(map + at, (size_t)sz));
				__atomic_store_n(state, (char)2, __ATOMIC_RELEASE);
				return title_cache[i];
			}
			std::this_thread::yield();
		}
	}
	// the letters of sequence i as stored (full_sseq)
	std::vector<int8_t> sequence(size_t i) const
	{
		if (!dmnd) return std::vector<int8_t>(all.data.begin() + all.limits[i], all.data.begin() + all.limits[i + 1] - 1);
		std::vector<int8_t> s(map + pos[i] + 1, map + pos[i] + 1 + len[i]);
		for (int8_t& l : s) l &= 31;
		return s;
	}
};

void write_dmnd(const std::string& path, const SeqBlock& b)
{
	std::string out = path;
	if (out.size() < 5 || out.substr(out.size() - 5) != ".dmnd") out += ".dmnd";
	std::ofstream f(out, std::ios::binary);
	if (!f) throw std::runtime_error("Error opening file " + out);
	const uint64_t n = b.ids.size();
	uint64_t magic = DMND_MAGIC, sequences = n, letters = (uint64_t)b.letters, pos_array_offset = 0;
	uint32_t build = 182, version = 3;
	auto header = [&] {
		f.seekp(0);
		f.write((char*)&magic, 8); f.write((char*)&build, 4); f.write((char*)&version, 4);
		f.write((char*)&sequences, 8); f.write((char*)&letters, 8); f.write((char*)&pos_array_offset, 8);
	};
	header();
	const uint64_t h2size = 48, zero = 0;
	f.write((char*)&h2size, 8);
	char hash[16] = { 0 };
	f.write(hash, 16);
	for (int i = 0; i < 4; ++i) f.write((char*)&zero, 8);
	uint64_t offset = (uint64_t)f.tellp();
	std::vector<uint64_t> pos;
	const char ff = (char)0xff;
	for (uint64_t i = 0; i < n; ++i) {
		const int64_t len = b.limits[i + 1] - b.limits[i] - 1;
		pos.push_back(offset);
		f.write(&ff, 1);
		f.write((const char*)b.data.data() + b.limits[i], len);
		f.write(&ff, 1);
		f.write(b.ids[i].c_str(), (std::streamsize)b.ids[i].size() + 1);
		offset += (uint64_t)len + b.ids[i].size() + 3;
	}
	pos_array_offset = offset;
	pos.push_back(offset);
	for (uint64_t i = 0; i <= n; ++i) {
		const uint32_t len = i < n ? (uint32_t)(b.limits[i + 1] - b.limits[i] - 1) : 0, pad = 0;
		f.write((char*)&pos[i], 8); f.write((char*)&len, 4); f.write((char*)&pad, 4);
	}
	header();
	std::cerr << "Database sequences  " << n << "\nDatabase letters  " << b.letters << "\n";
}

struct Options {
	std::string command, query, db, out, in;
	int threads = 0, k = 25, cbs = 1;
	double evalue = 0.001;
	bool fast = false;
	double block_size = 0.0;        // -b, billions of letters (0 = the reference's default for the sensitivity)
	int index_chunks = 0;           // -c (0 = the sensitivity's default)
	std::string masking = "", motif_masking = "", sens = "";
	int algo = -1;                  // --algo: -1 auto (the reference's default), 0 double-indexed, 1 query-indexed
	std::vector<std::string> outfmt;   // -f / --outfmt: format, then field names
	int gpus = 1;                   // --gpus: the reference blocks are spread over this many MI355X of the node
	double top = -1.0;              // --top PERCENT
	bool k_given = false;           // -k on the command line (--top excludes it, basic/config.cpp:674)
	int max_hsps = 1;               // --max-hsps N: HSPs per target (0 = all)
	int global_ranking = 0;         // --global-ranking N: extend only the N targets per query with the best ungapped scores over the whole database
	bool no_self_hits = false;      // --no-self-hits
	std::string matrix = "blosum62";        // --matrix / --gapopen / --gapextend (-1 = the matrix's default), basic/config.cpp:256-258
	int gap_open = -1, gap_extend = -1;
	std::string un, al;             // --un / --al: FASTA files of the queries without / with alignments
	int shapes = 0;                 // --shapes: the first N shapes of the sensitivity mode (ShapeConfig, basic/shape_config.h:34-44); 0 = all
	int ext = DMND_EXT_DEFAULT;     // --ext
	bool salltitles = false, sallseqid = false;      // DAA: full subject titles / all subject ids in the dictionary (DAAFormat, legacy/daa/daa_record.cpp:26-28)
	std::string daa;                // -a / --daa: the archive `view` reads; for blastp / blastx the legacy way to ask for DAA output into this file
	bool forwardonly = false;       // view: only alignments on the forward strand
	double dbsize = 0.0;            // --dbsize: effective database size in letters for the e-values (run/double_indexed.cpp:900); 0 = the database's
	int id2 = 0;                    // --id2: Hamming identities of a stage-1 hit (config.min_identities_, search/setup.cpp:343); 0 = the mode's
	double seed_cut = 0.0;          // --seed-cut: seed complexity cut (setup.cpp:368-369); 0 = the mode's
	double gapped_filter_evalue = -1.0;      // --gapped-filter-evalue (setup.cpp:346); -1 = the mode's, 0 = filter off
	int stop_match_score = 1;       // --stop-match-score: score of a stop codon against a stop codon (Scores ctor, stats/score_matrix.h:42-43)
	uint32_t format_flags = 0;      // --xml-blord-format, --no-parse-seqids, --sam-query-len
	bool compress = false;          // --compress 1: gzip output, ".gz" appended to the file name
	int frameshift = 0;             // -F / --frameshift: frame shift penalty, 0 = disabled (config.frame_shift)
	bool range_culling = false;     // --range-culling (config.query_range_culling); --long-reads = --range-culling --top 10 -F 15
	double range_cover = 50.0;      // --range-cover
	bool long_reads = false;
	int strands = 3, gencode = 1, min_orf = 0;      // --strand (mask: 1 plus, 2 minus), --query-gencode, --min-orf: translated searches
	int unal = -1;                  // --unal: report queries without alignments (-1 = the format's default)
	std::string header;             // --header [simple|verbose|0]
	double min_id = 0, query_cover = 0, subject_cover = 0, min_score = 0;      // --id, --query-cover, --subject-cover, --min-score
};

Options parse(int argc, char** argv)
{
	Options o;
	if (argc < 2) { o.command = "help"; return o; }
	o.command = argv[1];
	// short options may carry their value attached (-p4, -k3, -e10000, -b0.002, -c1), as the reference's parser accepts
	std::vector<std::string> args;
	for (int i = 2; i < argc; ++i) {
		const std::string a = argv[i];
		if (a.size() > 2 && a[0] == '-' && a[1] != '-' && std::string("pkebcqdofslat").find(a[1]) != std::string::npos) {
			args.push_back(a.substr(0, 2));
			args.push_back(a.substr(2));
		}
		else args.push_back(a);
	}
	const int n_args = (int)args.size();
	auto need = [&](int& i) -> std::string { if (i + 1 >= n_args) throw std::runtime_error("Missing parameter for option " + args[(size_t)i]); return args[(size_t)++i]; };
	for (int i = 0; i < n_args; ++i) {
		const std::string a = args[(size_t)i];
		if (a == "-q" || a == "--query") o.query = need(i);
		else if (a == "-d" || a == "--db") o.db = need(i);
		else if (a == "-o" || a == "--out") o.out = need(i);
		else if (a == "--in") o.in = need(i);
		else if (a == "-p" || a == "--threads") o.threads = std::atoi(need(i).c_str());
		else if (a == "-k" || a == "--max-target-seqs") {
			o.k = std::atoi(need(i).c_str());
			o.k_given = true;
			if (o.k < 0) throw std::runtime_error("Invalid value for --max-target-seqs.");
			if (o.k == 0) o.k = 1 << 30;                           // 0 = report every target (init_output, output/output_format.cpp:242-244)
		}
		else if (a == "-e" || a == "--evalue") o.evalue = std::atof(need(i).c_str());
		else if (a == "--fast") o.fast = true;
		else if (a == "--id") o.min_id = std::atof(need(i).c_str());
		else if (a == "--query-cover") o.query_cover = std::atof(need(i).c_str());
		else if (a == "--subject-cover") o.subject_cover = std::atof(need(i).c_str());
		else if (a == "--min-score") o.min_score = std::atof(need(i).c_str());
		else if (a == "--no-self-hits") o.no_self_hits = true;
		else if (a == "--matrix") o.matrix = need(i);
		else if (a == "-s" || a == "--shapes") { o.shapes = std::atoi(need(i).c_str()); if (o.shapes < 0) throw std::runtime_error("Invalid number of seed shapes."); }
		else if (a == "--ext") {
			const std::string v = need(i);                      // Extension::Mode, align/extend.cpp:51-58
			if (v == "banded-fast") o.ext = DMND_EXT_BANDED_FAST; else if (v == "banded-slow") o.ext = DMND_EXT_BANDED_SLOW; else if (v == "full") o.ext = DMND_EXT_FULL;
			else throw std::runtime_error("--ext " + v + " is not part of this build (banded-fast, banded-slow, full)");
		}
		else if (a == "--salltitles") o.salltitles = true;
		else if (a == "--sallseqid") o.sallseqid = true;
		else if (a == "-a" || a == "--daa") o.daa = need(i);
		else if (a == "--forwardonly") o.forwardonly = true;
		else if (a == "--dbsize") { o.dbsize = std::atof(need(i).c_str()); if (o.dbsize < 0) throw std::runtime_error("Invalid value for --dbsize."); }
		else if (a == "--id2") o.id2 = std::atoi(need(i).c_str());
		else if (a == "--seed-cut") o.seed_cut = std::atof(need(i).c_str());
		else if (a == "--gapped-filter-evalue") o.gapped_filter_evalue = std::atof(need(i).c_str());
		else if (a == "--stop-match-score") o.stop_match_score = std::atoi(need(i).c_str());
		else if (a == "--xml-blord-format") o.format_flags |= DMND_FMT_XML_BLORD;
		else if (a == "--no-parse-seqids") o.format_flags |= DMND_FMT_NO_PARSE_SEQIDS;
		else if (a == "--sam-query-len") o.format_flags |= DMND_FMT_SAM_QUERY_LEN;
		else if (a == "--un") o.un = need(i);
		else if (a == "--al") o.al = need(i);
		else if (a == "--unfmt" || a == "--alfmt") { if (need(i) != "fasta") throw std::runtime_error("Only the fasta format of --un / --al is part of this build."); }
		else if (a == "--compress") {
			const std::string v = need(i);                      // Config::compressor, basic/config.cpp:149-159
			if (v == "0") o.compress = false; else if (v == "1") o.compress = true;
			else throw std::runtime_error(v == "zstd" ? "Executable was not compiled with ZStd support." : "Invalid compression algorithm: " + v);
		}
		else if (a == "--strand") {
			const std::string v = need(i);                      // config.query_strands, basic/config.cpp:290,879
			if (v == "both") o.strands = 3; else if (v == "plus") o.strands = 1; else if (v == "minus") o.strands = 2;
			else throw std::runtime_error("Invalid value for parameter --strand");
		}
		else if (a == "--query-gencode") o.gencode = std::atoi(need(i).c_str());
		else if (a == "-l" || a == "--min-orf") o.min_orf = std::atoi(need(i).c_str());
		else if (a == "-F" || a == "--frameshift") { o.frameshift = std::atoi(need(i).c_str()); if (o.frameshift < 0) throw std::runtime_error("Invalid value for --frameshift."); }
		else if (a == "--range-culling") o.range_culling = true;
		else if (a == "--range-cover") o.range_cover = std::atof(need(i).c_str());
		else if (a == "--long-reads") o.long_reads = true;
		else if (a == "--gapopen") o.gap_open = std::atoi(need(i).c_str());
		else if (a == "--gapextend") o.gap_extend = std::atoi(need(i).c_str());
		else if (a == "--unal") { o.unal = std::atoi(need(i).c_str()); if (o.unal != 0 && o.unal != 1) throw std::runtime_error("Permitted values for --unal: 0, 1"); }
		else if (a == "--header") {
			o.header = "verbose";
			if (i + 1 < n_args && !args[(size_t)i + 1].empty() && args[(size_t)i + 1][0] != '-') o.header = args[(size_t)++i];
			if (o.header != "0" && o.header != "simple" && o.header != "verbose") throw std::runtime_error("Invalid header format: " + o.header);
		}
		else if (a == "--top") { o.top = std::atof(need(i).c_str()); if (o.top < 0.0 || o.top > 100.0) throw std::runtime_error("Invalid value for --top."); }
		else if (a == "--gpus") { o.gpus = std::atoi(need(i).c_str()); if (o.gpus < 1) throw std::runtime_error("Invalid number of GPUs."); }
		else if (a == "-b" || a == "--block-size") { o.block_size = std::atof(need(i).c_str()); if (o.block_size <= 0.0) throw std::runtime_error("Invalid block size."); }
		else if (a == "-c" || a == "--index-chunks") { o.index_chunks = std::atoi(need(i).c_str()); if (o.index_chunks < 1) throw std::runtime_error("Invalid number of index chunks."); }
		else if (a == "--comp-based-stats") { o.cbs = std::atoi(need(i).c_str()); if (o.cbs < 0 || o.cbs > 5) throw std::runtime_error("Invalid value for --comp-based-stats. Permitted values: 0, 1, 2, 3, 4, 5."); }
		else if (a == "--masking") o.masking = need(i);
		else if (a == "--motif-masking") o.motif_masking = need(i);
		else if (a == "--algo") {
			const std::string v = need(i);                      // Config::Algo: basic/config.cpp (0 | 1 | ctg, or the names)
			if (v == "0" || v == "double-indexed") o.algo = 0;
			else if (v == "1" || v == "query-indexed") o.algo = 1;
			else if (v == "auto" || v == "") o.algo = -1;
			else throw std::runtime_error("Invalid value for --algo: " + v + " (0 = double-indexed, 1 = query-indexed; ctg is not part of this build)");
		}
		else if (a == "-f" || a == "--outfmt") {
			o.outfmt.assign(1, need(i));
			while (i + 1 < n_args && !args[(size_t)i + 1].empty() && args[(size_t)i + 1][0] != '-') o.outfmt.push_back(args[(size_t)++i]);
		}
		else if (a == "--faster" || a == "--mid-sensitive" || a == "--sensitive" || a == "--more-sensitive" || a == "--very-sensitive" || a == "--ultra-sensitive")
			o.sens = a;
		else if (a == "--quiet" || a == "--log" || a == "-v" || a == "--verbose") {}
		else if (a == "-t" || a == "--tmpdir") (void)need(i);                // no temporary files: hits and records stay in memory / HBM
		else if (a == "--ignore-warnings" || a == "--no-auto-append" || a == "--keep-temp-files") {}
		else if (a == "--global-ranking" || a == "-g") { o.global_ranking = std::atoi(need(i).c_str()); if (o.global_ranking < 0) throw std::runtime_error("Invalid value for --global-ranking."); }
		else if (a == "--max-hsps") { o.max_hsps = std::atoi(need(i).c_str()); if (o.max_hsps < 0) throw std::runtime_error("Invalid value for --max-hsps."); }
		else if (a == "--custom-matrix") throw std::runtime_error("--custom-matrix is not part of this build (the standard matrices of --matrix are).");
		else if (a == "-g" || a == "--global-ranking" || a == "--swipe" || a == "--iterate" || a == "--approx-id" || a == "--taxonlist" || a == "--taxon-exclude" || a == "--seqidlist")
			throw std::runtime_error(a + " is not part of this build.");
		else throw std::runtime_error("Invalid option: " + a);
	}
	if (o.top >= 0.0 && o.k_given) throw std::runtime_error("--top and -k/--max-target-seqs are mutually exclusive.");      // basic/config.cpp:674-675
	if (o.long_reads) {                                       // basic/config.cpp:680-686
		o.range_culling = true;
		if (o.top < 0.0) o.top = 10.0;
		if (o.frameshift == 0) o.frameshift = 15;
	}
	return o;
}

// Output file: plain, or gzip-compressed with --compress 1 (OutputFile + ZlibSink, util/io/output_file.cpp:52-60)
struct Sink {
	FILE* f = nullptr;
	gzFile g = nullptr;
	void open(const std::string& path, bool gzip)
	{
		if (gzip) {
			if (path.empty()) throw std::runtime_error("--compress needs an output file (-o)");
			g = gzopen(path.c_str(), "wb");
			if (!g) throw std::runtime_error("Error opening file " + path);
			gzbuffer(g, 1 << 20);
		}
		else {
			f = path.empty() ? stdout : std::fopen(path.c_str(), "w");
			if (!f) throw std::runtime_error("Error opening file " + path);
		}
	}
	void write(const char* p, size_t n)
	{
		if (n == 0) return;
		if (g) { if (gzwrite(g, p, (unsigned)n) != (int)n) throw std::runtime_error("Error writing compressed output file"); }
		else if (std::fwrite(p, 1, n, f) != n) throw std::runtime_error("Error writing output file");
	}
	void write(const std::string& t) { write(t.data(), t.size()); }
	void close()
	{
		if (g) { if (gzclose(g) != Z_OK) throw std::runtime_error("Error closing compressed output file"); g = nullptr; }
		if (f && f != stdout) std::fclose(f);
		f = nullptr;
	}
};

double ms_since(std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); }

// DMND_CLI_TIMELINE=1: when every phase of a run ended, in ms since the process entered main (threads interleave; printed at the end)
struct Timeline {
	std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
	std::mutex m;
	std::vector<std::pair<double, std::string>> ev;
	const bool on = std::getenv("DMND_CLI_TIMELINE") != nullptr;
	void mark(const std::string& what) { if (!on) return; const double t = ms_since(t0); std::lock_guard<std::mutex> g(m); ev.emplace_back(t, what); }
	void print() { if (!on) return; std::sort(ev.begin(), ev.end()); for (auto& e : ev) std::fprintf(stderr, "timeline %9.2f ms  %s\n", e.first, e.second.c_str()); }
} g_timeline;

// Block boundaries of SequenceFile::load_seqs (src/data/sequence_file.cpp:215-222 for a .dmnd, :311-330 for FASTA):
// units (sequences, or reads with their six frames) are added while letters < max_letters, so a block ends with the unit
// that reaches the limit.
struct Range { size_t begin, end; };

std::vector<Range> split_blocks(const std::vector<int64_t>& unit_letters, int64_t max_letters)
{
	std::vector<Range> r;
	size_t i = 0;
	while (i < unit_letters.size()) {
		Range b{ i, i };
		int64_t letters = 0;
		do { letters += unit_letters[b.end++]; } while (b.end < unit_letters.size() && letters < max_letters);
		r.push_back(b);
		i = b.end;
	}
	return r;
}

// sequences [begin, end) of a loaded file as a block of their own (SequenceSet layout with its padding)
SeqBlock slice(const SeqBlock& all, size_t begin, size_t end)
{
	SeqBlock b;
	b.begin();
	b.data.insert(b.data.end(), all.data.begin() + all.limits[begin], all.data.begin() + all.limits[end]);
	for (size_t i = begin; i < end; ++i) b.limits.push_back(all.limits[i + 1] - all.limits[begin] + 256);
	b.letters = (all.limits[end] - all.limits[begin]) - (int64_t)(end - begin);
	b.finish();
	return b;
}

int run_blastp(const Options& o)
{
	if (o.db.empty()) throw std::runtime_error("Missing parameter: database file (--db/-d)");          // no -q: the queries come from standard input
	if (!o.sens.empty() && o.sens != "--sensitive" && o.sens != "--mid-sensitive" && o.sens != "--more-sensitive" && o.sens != "--very-sensitive" && o.sens != "--ultra-sensitive")
		throw std::runtime_error("This build implements --fast, default, --mid-sensitive, --sensitive, --more-sensitive, --very-sensitive and --ultra-sensitive (" + o.sens + " is not available).");
	if (o.fast && !o.sens.empty()) throw std::runtime_error("Conflicting sensitivity options.");
	// --masking (MaskingMode, run/config.cpp:124-135): tantan = default, on both blocks and on the GPU; seg = NCBI's SEG over the
	// reference block only, on the host as in the reference (dmnd_seg_mask_block); 0 / none
	const bool tantan = o.masking.empty() || o.masking == "1" || o.masking == "tantan";
	const bool seg = o.masking == "seg";
	if (!tantan && !seg && o.masking != "0" && o.masking != "none")
		throw std::runtime_error("Invalid value for --masking: " + o.masking + " (none / 0, seg, tantan / 1)");
	if (!o.motif_masking.empty() && o.motif_masking != "0" && o.motif_masking != "1") throw std::runtime_error("Permitted values for --motif-masking: 0, 1");
	// equal query and subject cover of 50 % and more switches the reference to its mutual-coverage search (length-sorted blocks, a
	// length-ratio cutoff inside the seed stage: run/config.cpp:156-159) -- a clustering path that is not part of this build
	if (o.command == "blastp" && o.query_cover >= 50 && o.query_cover == o.subject_cover)
		throw std::runtime_error("--query-cover equal to --subject-cover (>= 50) selects the reference's mutual-coverage search, which is not part of this build; use different values");
	// basic/config.cpp:688, :822-825: frameshift alignment is for translated queries, range culling for frameshift alignment
	if (o.global_ranking > 0 && (o.range_culling || o.frameshift > 0)) throw std::runtime_error("Global ranking is not supported in this mode.");
	if (o.frameshift != 0 && o.command == "blastp") throw std::runtime_error("Frameshift alignments are only supported for translated searches.");
	if (o.range_culling && o.frameshift == 0) throw std::runtime_error("Query range culling is only supported in frameshift alignment mode (option -F).");
	// basic/config.cpp:688, :700: what the matrix-adjust modes of --comp-based-stats exclude
	if (o.cbs >= 2 && o.global_ranking > 0) throw std::runtime_error("Global ranking is not supported in this mode.");
	if (o.cbs >= 2 && o.command == "blastx") throw std::runtime_error("This mode of composition based stats is not supported for translated searches.");
	const auto t_all = std::chrono::steady_clock::now();
	// the HIP runtime and the library's code object take their time to start (a quarter of a second on the MI355X boxes of this
	// project): that runs beside reading the queries, opening the database and loading the first reference block
	// (the error text is thread-local in the library: it travels back with the code)
	std::future<std::pair<int, std::string>> gpu_ready = std::async(std::launch::async, [] {
		const int rc = dmnd_init(-1);
		g_timeline.mark("dmnd_init");
		return std::make_pair(rc, std::string(rc == DMND_OK ? "" : dmnd_last_error()));
	});
	// tantan's likelihood ratios need the scoring matrix' lambda, a 6 ms root search (kept per matrix by the library): beside the file reads too
	std::future<void> lambda_ready = std::async(std::launch::async, [&o, tantan] {
		dmnd_params mp;
		dmnd_default_params(&mp);
		if (tantan && dmnd_matrix_params(o.matrix.c_str(), o.gap_open, o.gap_extend, &mp) == DMND_OK) (void)dmnd_masking_lambda(&mp);
	});
	SeqBlock q_all;
	const bool blastx = o.command == "blastx";
	const size_t C = blastx ? 6 : 1;
	std::vector<int32_t> source_len;
	std::vector<std::string> read_ids;
	std::vector<std::vector<int8_t>> reads;
	// --outfmt (output/output_format.cpp:178-200): 6 / tab with optional field names, 0 / pairwise
	enum { FMT_TAB, FMT_FIELDS, FMT_PAIRWISE, FMT_PAF, FMT_SAM, FMT_XML, FMT_DAA } fmt = FMT_TAB;
	std::vector<int32_t> field_ids;
	int need_transcripts = 0;
	if (!o.outfmt.empty()) {
		const std::string& f0 = o.outfmt[0];
		if (f0 == "6" || f0 == "tab") {
			if (o.outfmt.size() > 1) {
				std::vector<const char*> names;
				for (size_t i = 1; i < o.outfmt.size(); ++i) names.push_back(o.outfmt[i].c_str());
				field_ids.resize(names.size());
				if (dmnd_output_fields(names.data(), (int)names.size(), field_ids.data(), &need_transcripts) != DMND_OK) throw std::runtime_error(dmnd_last_error());
				fmt = FMT_FIELDS;
			}
		}
		else if (f0 == "0" || f0 == "pairwise") {
			if (o.outfmt.size() > 1) throw std::runtime_error("Invalid output format: the pairwise format takes no fields");
			fmt = FMT_PAIRWISE;
			need_transcripts = 1;
		}
		else if (f0 == "103" || f0 == "paf") {
			if (o.outfmt.size() > 1) throw std::runtime_error("Invalid output format: the PAF format takes no fields");
			fmt = FMT_PAF;
		}
		else if (f0 == "101" || f0 == "sam") {
			if (o.outfmt.size() > 1) throw std::runtime_error("Invalid output format: the SAM format takes no fields");
			fmt = FMT_SAM;
			need_transcripts = 1;
		}
		else if (f0 == "100" || f0 == "daa") {
			if (o.outfmt.size() > 1) throw std::runtime_error("Invalid output format: the DAA format takes no fields");
			if (o.out.empty()) throw std::runtime_error("The DAA format needs an output file (-o)");
			if (o.compress) throw std::runtime_error("Compression is not supported for DAA format.");
			fmt = FMT_DAA;
			need_transcripts = 1;
		}
		else if (f0 == "5" || f0 == "xml") {
			if (o.outfmt.size() > 1) throw std::runtime_error("Invalid output format: the XML format takes no fields");
			fmt = FMT_XML;
			need_transcripts = 1;
		}
		else throw std::runtime_error("Invalid output format: " + f0 + " (this build prints 6 = BLAST tabular, 0 = BLAST pairwise, 5 = BLAST XML, 100 = DAA, 101 = SAM and 103 = PAF)");
	}
	// --unal / --header work on the field list: the default columns are a field list too
	const bool tab_extras = o.unal == 1 || (!o.header.empty() && o.header != "0");
	if (fmt == FMT_TAB && tab_extras) {
		static const char* const std_fields[12] = { "qseqid", "sseqid", "pident", "length", "mismatch", "gapopen", "qstart", "qend", "sstart", "send", "evalue", "bitscore" };
		field_ids.resize(12);
		if (dmnd_output_fields(std_fields, 12, field_ids.data(), &need_transcripts) != DMND_OK) throw std::runtime_error(dmnd_last_error());
		fmt = FMT_FIELDS;
	}
	if (tab_extras && fmt != FMT_FIELDS && !o.header.empty() && o.header != "0") throw std::runtime_error("--header is only available for the tabular format");
	if (o.frameshift > 0) {
		need_transcripts = 1;      // config.frame_shift != 0 => output_format->hsp_values = TRANSCRIPT (output/output_format.cpp:256-257); the block join re-counts from them
		// A frameshift alignment changes frame along its transcript: the cursor of format_api.hip follows the frames, and the tabular
		// format prints such alignments with any field. The other writers print them too (`view` of a reference-written -F archive
		// equals the reference's view in the pairwise, XML, SAM and PAF formats: tests/test_view.py), but which queries WITHOUT an
		// alignment those formats list differs from the reference under -F (its legacy pipeline reports by its own rule), so the
		// search itself stays with the tabular format.
		// A DAA archive never lists unaligned queries, and its records are written from the first column's frame and position
		// (`view` of a reference-written -F archive writes every record back byte for byte: tests/test_view.py), so -f 100 is fine.
		if (fmt == FMT_SAM) throw std::runtime_error("Frameshift alignments (-F): the SAM format is not available in this build (its query column reads the alignment's range from one frame, past that frame's end in the reference); use -f 100 and view.");
		if (dmnd_set_format_flags(o.format_flags | DMND_FMT_FRAMESHIFT) != DMND_OK) throw std::runtime_error(dmnd_last_error());      // qseq_translated follows the alignment (config.frame_shift != 0)
	}
	// which formats report queries without alignments: pairwise, PAF and SAM by default (DEFAULT_REPORT_UNALIGNED), tabular with --unal 1
	// (DAA files never list unaligned queries, whatever --unal says: output/join_blocks.cpp:302,365)
	const bool report_unal = fmt != FMT_DAA && (o.unal == 1 || (o.unal == -1 && (fmt == FMT_PAIRWISE || fmt == FMT_PAF || fmt == FMT_SAM || fmt == FMT_XML)));
	bool want_full_sseq = false;
	for (int32_t id : field_ids) want_full_sseq |= id == DMND_F_FULL_SSEQ;
	auto t0 = std::chrono::steady_clock::now();
	// with a frameshift penalty every open reading frame counts, however short (Config::min_orf_len, basic/config.h:413-421)
	const int min_orf = o.min_orf > 0 ? o.min_orf : o.frameshift != 0 ? 1 : 0;
	if (blastx) read_dna_fasta_translated(o.query, q_all, source_len, read_ids, reads, o.gencode, o.strands, min_orf);
	else read_fasta(o.query, q_all);
	std::string dbpath = o.db;
	if (!std::ifstream(dbpath).good() && std::ifstream(dbpath + ".dmnd").good()) dbpath += ".dmnd";
	Database db;
	db.open(dbpath);
	const size_t n_queries = q_all.ids.size() / C, n_targets = db.n;
	std::cerr << "Loading sequences...  [" << ms_since(t0) / 1e3 << "s]  queries=" << n_queries << " targets=" << n_targets << " letters=" << db.letters << "\n";
	g_timeline.mark("queries read, database opened");
	const int sens = o.fast ? DMND_SENS_FAST : o.sens == "--mid-sensitive" ? DMND_SENS_MID_SENSITIVE : o.sens == "--sensitive" ? DMND_SENS_SENSITIVE
		: o.sens == "--more-sensitive" ? DMND_SENS_MORE_SENSITIVE : o.sens == "--very-sensitive" ? DMND_SENS_VERY_SENSITIVE
		: o.sens == "--ultra-sensitive" ? DMND_SENS_ULTRA_SENSITIVE : DMND_SENS_DEFAULT;
	// -b: 2.0 billion letters, 0.4 from --very-sensitive up (run/double_indexed.cpp:792-795); basic/config.h:434
	const double b_opt = o.block_size > 0.0 ? o.block_size : (sens >= DMND_SENS_VERY_SENSITIVE ? 0.4 : 2.0);
	const int64_t max_letters = (int64_t)(b_opt * 1e9);
	std::vector<int64_t> q_units(n_queries), t_units(n_targets);
	for (size_t i = 0; i < n_targets; ++i) t_units[i] = db.length(i);
	for (size_t i = 0; i < n_queries; ++i) {
		if (!blastx) { q_units[i] = q_all.limits[i + 1] - q_all.limits[i] - 1; continue; }
		// Block::push_back counts the letters of the ORFs that survive find_orfs (data/block/block.cpp:88-100)
		const int l0 = (int)(q_all.limits[i * 6 + 1] - q_all.limits[i * 6] - 1), min_len = min_orf > 0 ? min_orf : l0 < 30 ? 1 : l0 < 100 ? 20 : 40;
		int64_t n = 0;
		for (size_t f = 0; f < 6; ++f) {
			if (!(o.strands & (f < 3 ? 1 : 2))) continue;             // frames of a strand that is not searched hold no ORF letters (block.cpp:92-99)
			const int8_t* s = q_all.data.data() + q_all.limits[i * 6 + f];
			const int len = (int)(q_all.limits[i * 6 + f + 1] - q_all.limits[i * 6 + f] - 1);
			for (int x = 0, begin = 0; x <= len; ++x)
				if (x == len || s[x] == 24) { if (x - begin >= min_len) n += x - begin; begin = x + 1; }
		}
		q_units[i] = n;
	}
	// --gpus N: the reference blocks are the unit of distribution (a block's seed stage and extension are independent of the
	// other blocks', and the records of a query are merged over the blocks as join_blocks does), so the database is cut into at
	// least N of them -- the block size is lowered to total letters / N if -b would give fewer. The output is the reference's
	// for the same block cut (-b).
	const int n_gpus = o.gpus;
	// test hook (a box with one GPU): DMND_CLI_SHARE_GPU=1 runs the N host threads and contexts of --gpus N on the current device
	const bool share_gpu = std::getenv("DMND_CLI_SHARE_GPU") != nullptr && std::getenv("DMND_CLI_SHARE_GPU")[0] == '1';
	if (!share_gpu) {
		const int have = dmnd_device_count();
		if (n_gpus > have) throw std::runtime_error("--gpus " + std::to_string(n_gpus) + ": only " + std::to_string(have) + " gfx950 device(s) visible");
	}
	int64_t t_max_letters = max_letters;
	if (n_gpus > 1) t_max_letters = std::min<int64_t>(max_letters, (db.letters + n_gpus - 1) / n_gpus);
	const std::vector<Range> q_blocks = split_blocks(q_units, max_letters), t_blocks = split_blocks(t_units, t_max_letters);
	if (q_blocks.size() > 1 || t_blocks.size() > 1)
		std::cerr << "Block size = " << max_letters << "  query blocks=" << q_blocks.size() << " reference blocks=" << t_blocks.size() << "\n";
	const int load_threads = o.threads > 0 ? o.threads : 8;
	// every GPU's first reference block is read (mapped file -> SequenceSet layout) while the contexts are being created
	std::vector<std::future<SeqBlock>> first_block((size_t)n_gpus);
	for (int g = 0; g < n_gpus && (size_t)g < t_blocks.size(); ++g)
		first_block[(size_t)g] = std::async(std::launch::async, [&db, &t_blocks, g, load_threads] {
			SeqBlock b = db.load(t_blocks[(size_t)g].begin, t_blocks[(size_t)g].end, load_threads);
			g_timeline.mark("first reference block of GPU " + std::to_string(g) + " loaded");
			return b;
		});

	dmnd_params p;
	dmnd_default_params(&p);
	if (dmnd_matrix_params(o.matrix.c_str(), o.gap_open, o.gap_extend, &p) != DMND_OK) throw std::runtime_error(dmnd_last_error());
	p.db_letters = o.dbsize > 0.0 ? o.dbsize : (double)db.letters;
	if (o.stop_match_score != 1) p.matrix8[24 * 32 + 24] = (int8_t)o.stop_match_score;
	p.max_evalue = o.evalue;
	auto chk = [&](int rc) { if (rc != DMND_OK) throw std::runtime_error(dmnd_last_error()); };
	const int threads = o.threads > 0 ? o.threads : 8;
	dmnd_seed_params sp;
	double gf_evalue = 0.0;
	chk(dmnd_seed_params_preset(&sp, sens, threads, &p, &gf_evalue));
	if (o.gapped_filter_evalue >= 0.0) gf_evalue = o.gapped_filter_evalue;
	if (o.id2 > 0) sp.hamming_filter_id = o.id2;
	if (o.seed_cut != 0.0) sp.seed_complexity_cut = o.seed_cut * 0.69314718055994530942 * sp.shape_weight[0];
	if (o.index_chunks > 0) chk(dmnd_seed_params_set_index_chunks(&sp, o.index_chunks, threads));
	if (o.shapes > 0 && o.shapes < sp.n_shapes) sp.n_shapes = o.shapes;
	// one context per GPU, driven by its own host thread (measured, round 4: creating the context and starting the first block's upload
	// while dmnd_init still loads code objects stretches that load from 95 to 123 ms -- page-locking 300 MB and the loader contend --
	// and the seed stage starts 6 ms later than with the wait here)
	{ const std::pair<int, std::string> r = gpu_ready.get(); if (r.first != DMND_OK) throw std::runtime_error(r.second.empty() ? "dmnd_init failed" : r.second); }
	std::vector<dmnd_ctx*> ctxs((size_t)n_gpus, nullptr);
	for (int g = 0; g < n_gpus; ++g) {
		dmnd_ctx* c = dmnd_create(n_gpus > 1 && !share_gpu ? g : -1, &p);
		if (!c) throw std::runtime_error(dmnd_last_error());
		ctxs[(size_t)g] = c;
		g_timeline.mark("context " + std::to_string(g) + " created");
		chk(dmnd_set_max_target_seqs(c, o.k));
		chk(dmnd_set_max_hsps(c, o.max_hsps));
		chk(dmnd_set_top_percent(c, o.top));
		chk(dmnd_set_filters(c, o.min_id, o.query_cover, o.subject_cover, o.min_score));
		chk(dmnd_set_comp_based_stats(c, o.cbs));
		chk(dmnd_set_query_contexts(c, blastx ? 6 : 1));
		chk(dmnd_set_frameshift(c, o.frameshift, o.range_culling ? 1 : 0, o.range_cover, 16));
		chk(dmnd_set_sensitivity(c, sens));
		// global ranking extends its targets over the full matrix (search/setup.cpp:377-389)
		if (o.global_ranking > 0 && o.ext != DMND_EXT_DEFAULT && o.ext != DMND_EXT_FULL) throw std::runtime_error("Global ranking only supports full matrix extension.");
		chk(dmnd_set_extension_mode(c, o.global_ranking > 0 ? DMND_EXT_FULL : o.ext));
		chk(dmnd_set_global_ranking(c, o.global_ranking));
		chk(dmnd_set_gapped_filter(c, gf_evalue));
		// several reference blocks per GPU: the query seed index of a query block is built once and kept for all of them
		if (t_blocks.size() > (size_t)n_gpus) chk(dmnd_set_query_index_reuse(c, 1));
	}
	sp.query_translated = blastx ? 1 : 0;
	// --algo (run/double_indexed.cpp:267-300): auto = query-indexed for a query block of at most 32 Mi letters against a
	// database of 256 MiB and more (the size of the file on disk), decided on the first query block as the reference does
	int algo = o.algo;
	if (algo < 0) {
		int64_t db_bytes = 0;
		{ std::ifstream f(dbpath, std::ios::binary | std::ios::ate); if (f) db_bytes = (int64_t)f.tellg(); }
		const Range& q0 = q_blocks.front();
		std::vector<int64_t> lim(q_all.limits.begin() + (ptrdiff_t)(q0.begin * C), q_all.limits.begin() + (ptrdiff_t)(q0.end * C) + 1);
		int qi = 0;
		chk(dmnd_auto_query_indexed(&sp, q_all.data.data(), lim.data(), (int64_t)lim.size() - 1, db_bytes, &qi));
		algo = qi;
	}
	std::cerr << "Algorithm: " << (algo == 1 ? "Query-indexed" : "Double-indexed") << "\n";
	if (algo == 1) {
		if (o.index_chunks > 1) throw std::runtime_error("The query-indexed algorithm of this build runs with one index chunk (-c1).");
		chk(dmnd_seed_params_set_query_indexed(&sp, threads));
	}
	// motif soft masking (soft_masking_algo, search/setup.cpp:322-335): on by default up to --sensitive, needs masking enabled
	// when forced on. The motif table is reference data (tools/make_motif_table.py -> motifs.bin next to this binary).
	bool motifs = o.motif_masking.empty() ? sens <= DMND_SENS_SENSITIVE : o.motif_masking == "1";
	if (o.motif_masking == "1" && !tantan && !seg) throw std::runtime_error("Soft masking requires masking.");
	if (motifs) {
		std::string dir = ".";
		{ char buf[4096]; const ssize_t n = readlink("/proc/self/exe", buf, sizeof buf - 1); if (n > 0) { buf[n] = 0; dir = buf; dir = dir.substr(0, dir.find_last_of('/')); } }
		std::ifstream f(dir + "/motifs.bin", std::ios::binary);
		std::vector<uint64_t> codes;
		uint64_t c;
		while (f.read((char*)&c, 8)) codes.push_back(c);
		// without the table the default command line would silently differ from the reference's: that is an error, not a warning
		if (codes.empty()) throw std::runtime_error("Motif masking is on (the default up to --sensitive) but " + dir + "/motifs.bin was not found: generate it with tools/make_motif_table.py, or pass --motif-masking 0.");
		chk(dmnd_set_motif_table(codes.data(), (int64_t)codes.size()));
	}
	// query-indexed + masking: the reference masks a target only when the extension stage loads it (lazy masking,
	// extend.cpp:168-181; run/double_indexed.cpp:300), i.e. the seed stage sees the unmasked reference block
	const bool lazy_masking = algo == 1 && (tantan || seg) && o.frameshift == 0;

	Sink out;
	{
		std::string path = o.out;                // auto_append_extension(output_file, ".gz"), basic/config.cpp:771-772
		if (o.compress && (path.size() < 3 || path.substr(path.size() - 3) != ".gz")) path += ".gz";
		if (fmt == FMT_DAA && (path.size() < 4 || path.substr(path.size() - 4) != ".daa")) path += ".daa";      // auto_append_extension, config.cpp:728
		out.open(path, o.compress);
	}
	// DAA (legacy/daa/daa_write.cpp): header placeholder now, query records as they come, dictionary + final header at the end
	dmnd_daa_header daa;
	std::vector<uint32_t> daa_dict_id;                 // target ordinal -> dictionary id (order of first appearance in the output)
	std::vector<uint32_t> daa_dict;                    // dictionary id -> target ordinal
	int64_t daa_bytes = 0, daa_queries = 0;
	if (fmt == FMT_DAA) {
		daa.build = 182;                                // the build number of the reference version whose format this is (basic/const.h:25)
		daa.db_seqs = (int64_t)db.n; daa.db_letters = db.letters;      /* the database's letters, not --dbsize */ daa.db_seqs_used = 0; daa.query_records = 0;
		daa.mode = blastx ? 3 : 2; daa.gap_open = p.gap_open; daa.gap_extend = p.gap_extend; daa.K = p.K; daa.lambda = p.lambda; daa.max_evalue = o.evalue;
		daa.matrix = o.matrix.c_str(); daa.finished = 0; daa.alignment_bytes = 0; daa.ref_name_bytes = 0;
		std::vector<char> hb(4096);
		const int64_t w = dmnd_format_daa_header(&daa, hb.data(), (int64_t)hb.size());
		if (w < 0) throw std::runtime_error(dmnd_last_error());
		out.write(hb.data(), (size_t)w);
		daa_dict_id.assign(db.n, UINT32_MAX);
	}
	if (fmt == FMT_PAIRWISE) out.write("BLASTP 2.3.0+\n\n\n");                   // PairwiseFormat::print_header
	if (fmt == FMT_FIELDS && o.header == "simple") {
		std::vector<char> hb(4096);
		const int64_t w = dmnd_format_fields_header(field_ids.data(), (int)field_ids.size(), hb.data(), (int64_t)hb.size());
		if (w < 0) throw std::runtime_error(dmnd_last_error());
		out.write(hb.data(), (size_t)w);
	}
	if (fmt == FMT_FIELDS && o.header == "verbose") {       // TabularFormat::print_header (program name and command line are ours)
		std::string h = "# diamond-hip (ABI " + std::to_string(dmnd_abi_version()) + "). MI355X back end of DIAMOND's seed-and-extend path\n# Fields: ";
		for (size_t i = 1; i < o.outfmt.size(); ++i) h += (i > 1 ? ", " : "") + o.outfmt[i];
		if (o.outfmt.size() <= 1) h += "qseqid, sseqid, pident, length, mismatch, gapopen, qstart, qend, sstart, send, evalue, bitscore";
		out.write(h + "\n");
	}
	if (fmt == FMT_SAM) {                                                              // SamFormat::print_header (program name and version are ours)
		const std::string prog = blastx ? "BlastX" : "BlastP";
		out.write("@HD\tVN:1.5\tSO:query\n@PG\tPN:diamond-hip\tVN:ABI" + std::to_string(dmnd_abi_version()) + "\n@mm\t" + prog + "\n@CO\t" + prog
			+ "-like alignments\n@CO\tReporting AS: bitScore, ZR: rawScore, ZE: expected, ZI: percent identity, ZL: reference length, ZF: frame, ZS: query start DNA coordinate\n");
	}
	if (fmt == FMT_XML) {                                                              // XMLFormat::print_header (version string is ours)
		const std::vector<std::string>& titles = blastx ? read_ids : q_all.ids;
		if (!titles.empty()) {
			const int32_t len0 = blastx ? source_len[0] : (int32_t)(q_all.limits[1] - q_all.limits[0] - 1);
			std::vector<char> hb(titles[0].size() * 6 + o.db.size() + 4096);
			const int64_t w = dmnd_format_xml_header(blastx ? "blastx" : "blastp", ("diamond-hip ABI " + std::to_string(dmnd_abi_version())).c_str(), o.db.c_str(), titles[0].c_str(), len0,
				o.matrix.c_str(), p.gap_open, p.gap_extend, o.evalue, hb.data(), (int64_t)hb.size());
			if (w < 0) throw std::runtime_error(dmnd_last_error());
			out.write(hb.data(), (size_t)w);
		}
	}
	FILE* un_file = nullptr; FILE* al_file = nullptr;          // --un / --al: config.unaligned / aligned_file, run/double_indexed.cpp:689-693
	if (!o.un.empty() && !(un_file = std::fopen(o.un.c_str(), "w"))) throw std::runtime_error("Error opening file " + o.un);
	if (!o.al.empty() && !(al_file = std::fopen(o.al.c_str(), "w"))) throw std::runtime_error("Error opening file " + o.al);
	const std::vector<std::string>& qtitles = blastx ? read_ids : q_all.ids;
	std::vector<std::string> qid(qtitles.size());
	for (size_t i = 0; i < qid.size(); ++i) qid[i] = short_id(qtitles[i]);
	double ms_upload = 0, ms_mask = 0, ms_seed = 0, ms_ext = 0;       // summed over the GPUs' host threads
	int64_t motif_letters = 0;
	int64_t total_hits = 0, total_matches = 0, aligned = 0, mq_total = 0, mt_total = 0;
	char line[8192];
	std::mutex merge_mutex;
	std::vector<std::vector<int8_t>> t_masked((size_t)n_gpus);      // lazily masked copy of the reference block at hand (query-indexed algorithm)
	// --no-self-hits: the library finds query / target pairs with the same letters and asks here whether the titles agree too
	struct SelfCtx { const std::vector<std::string>* qtitles; const Database* db; size_t q0 = 0, t0 = 0; const std::vector<uint32_t>* ordinals = nullptr; };      // ordinals: block id -> database ordinal (globally ranked targets)
	std::vector<SelfCtx> self_ctx((size_t)n_gpus);
	struct Held { SeqBlock block; size_t index = (size_t)-1; bool ahead = false; };      // ahead: read AND uploaded before the query phase, not yet masked
	std::vector<Held> held_blocks((size_t)n_gpus);                  // the reference block every GPU's thread holds in host memory
	// f(g) on one host thread per GPU; the first error is rethrown on the calling thread
	auto on_each_gpu = [&](const std::function<void(int)>& f) {
		if (n_gpus == 1) { f(0); return; }
		std::vector<std::thread> th;
		std::vector<std::string> err((size_t)n_gpus);
		for (int g = 0; g < n_gpus; ++g)
			th.emplace_back([&, g] { try { f(g); } catch (const std::exception& e) { err[(size_t)g] = e.what()[0] ? e.what() : "error"; } });
		for (auto& t : th) t.join();
		for (const std::string& e : err) if (!e.empty()) throw std::runtime_error(e);
	};
	for (const Range& qr : q_blocks) {
		SeqBlock q_own;
		if (q_blocks.size() > 1) q_own = slice(q_all, qr.begin * C, qr.end * C);
		SeqBlock& q = q_blocks.size() > 1 ? q_own : q_all;
		const int64_t nq = (int64_t)((qr.end - qr.begin) * C);
		// every GPU gets the query block; GPU 0's masking run also writes the masked letters into the host copy that the host part
		// of every extension call reads. Two phases with a join between them: no GPU's upload may still be reading the host copy
		// when GPU 0 writes the masked letters back into it (a late GPU would upload a partly masked block and mask it again).
		std::vector<double> up_ms((size_t)n_gpus, 0.0);
		// first query block: every GPU's first reference block goes to HBM on a helper thread meanwhile (the reference block has its
		// own transfer lane in the library); SEG masks the host copy before the upload, so not then
		std::vector<std::future<double>> ahead((size_t)n_gpus);
		if (&qr == &q_blocks.front() && !seg && !std::getenv("DMND_CLI_NO_AHEAD"))
			for (int g = 0; g < n_gpus && (size_t)g < t_blocks.size(); ++g)
				ahead[(size_t)g] = std::async(std::launch::async, [&, g]() -> double {
					Held& h = held_blocks[(size_t)g];
					h.block = first_block[(size_t)g].get();
					h.index = (size_t)g;
					g_timeline.mark("reference block " + std::to_string(g) + " in host memory");
					const auto t0 = std::chrono::steady_clock::now();
					const Range& tr = t_blocks[(size_t)g];
					if (dmnd_upload_block(ctxs[(size_t)g], DMND_TARGET, h.block.data.data(), (int64_t)h.block.data.size(), h.block.limits.data(), (int64_t)(tr.end - tr.begin)) != DMND_OK)
						throw std::runtime_error(dmnd_last_error());
					h.ahead = true;
					g_timeline.mark("reference block " + std::to_string(g) + " uploaded (beside the query phase)");
					return ms_since(t0);
				});
		on_each_gpu([&](int g) {
			auto t0 = std::chrono::steady_clock::now();
			chk(dmnd_upload_block(ctxs[(size_t)g], DMND_QUERY, q.data.data(), (int64_t)q.data.size(), q.limits.data(), nq));
			up_ms[(size_t)g] = ms_since(t0);
		});
		on_each_gpu([&](int g) {
			dmnd_ctx* ctx = ctxs[(size_t)g];
			const double up = up_ms[(size_t)g];
			auto t0 = std::chrono::steady_clock::now();
			int64_t mq = 0, ml = 0;
			g_timeline.mark("query block uploaded");
			if (tantan) chk(dmnd_mask_block(ctx, DMND_QUERY, g == 0 ? q.data.data() : nullptr, &mq));
			const double mk = ms_since(t0);
			if (tantan) g_timeline.mark("query block masked (tantan)");
			if (motifs) chk(dmnd_soft_mask_block(ctx, DMND_QUERY, &ml));
			if (blastx) chk(dmnd_set_query_source_lengths(ctx, source_len.data() + qr.begin, (int64_t)(qr.end - qr.begin)));
			std::lock_guard<std::mutex> lock(merge_mutex);
			ms_upload += up; ms_mask += mk;
			if (g == 0) { mq_total += mq; motif_letters += ml; }
		});
		g_timeline.mark("query block uploaded and masked");
		// the seed stage's device buffers are allocated beside the upload and the masking of the first reference block
		std::vector<std::future<int>> reserved;
		if (&qr == &q_blocks.front())
			for (int g = 0; g < n_gpus; ++g)
				reserved.push_back(std::async(std::launch::async, [&, g] { int rc = dmnd_seed_reserve(ctxs[(size_t)g], &sp, (int64_t)q.data.size()); if (rc == DMND_OK) rc = dmnd_extend_reserve(ctxs[(size_t)g], (int64_t)(4 * (qr.end - qr.begin))); g_timeline.mark("seed + extension buffers reserved"); return rc; }));
		// --global-ranking: per query the N best targets of the whole database by ungapped score (align/global_ranking/table.cpp)
		std::vector<dmnd_ranked_target> rank_table(o.global_ranking > 0 ? (qr.end - qr.begin) * (size_t)o.global_ranking : 0, dmnd_ranked_target{ 0, 0, 0, 0, 0 });
		std::vector<dmnd_match> joined;                       // the query block's records against all reference blocks
		std::vector<uint8_t> joined_gpu;                      // ... and the GPU that produced each of them (the RCCL merge sends from there)
		std::vector<uint8_t> arena;                           // ... and their transcripts, if the output format reads them
		std::vector<char> seeded(qr.end - qr.begin, 0);       // queries with at least one seed hit (what the unaligned report depends on)
		on_each_gpu([&](int g) {
		dmnd_ctx* ctx = ctxs[(size_t)g];
		Held& held = held_blocks[(size_t)g];
		std::future<SeqBlock> next;                           // the GPU's next reference block, read while this one is searched
		for (size_t bi = (size_t)g; bi < t_blocks.size(); bi += (size_t)n_gpus) {
			const Range& tr = t_blocks[bi];
			// the reference re-reads and re-masks every reference block for every query block (run/double_indexed.cpp:404-470); a
			// GPU that has one block keeps it (masked) from one query block to the next
			double ahead_ms = 0;
			if ((size_t)g < ahead.size() && ahead[(size_t)g].valid()) ahead_ms = ahead[(size_t)g].get();      // rethrows the helper's error
			const bool in_hbm = held.ahead && held.index == bi;
			held.ahead = false;
			const bool fresh = held.index != bi || in_hbm;   // just read: unmasked
			if (fresh && !in_hbm) {
				held.block = bi == (size_t)g && first_block[(size_t)g].valid() ? first_block[(size_t)g].get() : next.valid() ? next.get() : db.load(tr.begin, tr.end, threads);
				held.index = bi;
				g_timeline.mark("reference block " + std::to_string(bi) + " in host memory");
			}
			const size_t bn = bi + (size_t)n_gpus;
			if (bn < t_blocks.size()) next = std::async(std::launch::async, [&db, &t_blocks, bn, threads] { return db.load(t_blocks[bn].begin, t_blocks[bn].end, threads); });
			SeqBlock& t = held.block;
			auto t0 = std::chrono::steady_clock::now();
			double up = 0, mk = 0;
			int64_t mt = 0, ml = 0;
			const int64_t t_seqs = (int64_t)(tr.end - tr.begin);
			// SEG runs on the host copy: a block that was just read is masked before it goes to HBM (up-front masking)
			if (seg && !lazy_masking && fresh) { chk(dmnd_seg_mask_block(t.data.data(), t.limits.data(), t_seqs, threads, &mt)); mk += ms_since(t0); t0 = std::chrono::steady_clock::now(); }
			if (in_hbm) up += ahead_ms;
			else {
				chk(dmnd_upload_block(ctx, DMND_TARGET, t.data.data(), (int64_t)t.data.size(), t.limits.data(), t_seqs));
				up += ms_since(t0);
				g_timeline.mark("reference block " + std::to_string(bi) + " uploaded");
			}
			const int8_t* t_host = t.data.data();            // the letters the extension stage's host part reads
			std::vector<int32_t> lazy_ids;                   // lazy masking: the targets that have seed hits (block sequence ids)
			auto mask_target = [&] {
				t0 = std::chrono::steady_clock::now();
				if (lazy_masking && q_blocks.size() == 1 && !seg) {
					// the only query block: nobody needs the unmasked letters again, the host copy is masked in place -- and, as in
					// the reference (extend.cpp:168-181), only the targets the extension stage will load
					chk(dmnd_mask_sequences(ctx, DMND_TARGET, t.data.data(), lazy_ids.data(), (int64_t)lazy_ids.size(), &mt));
				}
				else if (lazy_masking) {                       // t.data stays unmasked: the next query block's seed stage needs it so
					t_masked[(size_t)g].resize(t.data.size());
					std::memcpy(t_masked[(size_t)g].data(), t.data.data(), t.data.size());
					if (seg) {                                   // masked copy on the host, then the block in HBM is replaced by it
						chk(dmnd_seg_mask_block(t_masked[(size_t)g].data(), t.limits.data(), t_seqs, threads, &mt));
						chk(dmnd_upload_block(ctx, DMND_TARGET, t_masked[(size_t)g].data(), (int64_t)t.data.size(), t.limits.data(), t_seqs));
					}
					else chk(dmnd_mask_sequences(ctx, DMND_TARGET, t_masked[(size_t)g].data(), lazy_ids.data(), (int64_t)lazy_ids.size(), &mt));      // patches the masked positions into the copy
					t_host = t_masked[(size_t)g].data();
				}
				else chk(dmnd_mask_block(ctx, DMND_TARGET, t.data.data(), &mt));
				mk += ms_since(t0);
			};
			// up-front masking of a block that was just read; a block the GPU kept from the previous query block carries its masked
			// letters already and is uploaded as it is
			if (tantan && !lazy_masking && fresh) { mask_target(); g_timeline.mark("reference block " + std::to_string(bi) + " masked (tantan)"); }
			if (motifs && algo == 0) { chk(dmnd_soft_mask_block(ctx, DMND_TARGET, &ml)); g_timeline.mark("reference block " + std::to_string(bi) + " soft-masked (motifs)"); }
			if ((size_t)g < reserved.size() && reserved[(size_t)g].valid()) (void)reserved[(size_t)g].get();      // an optimisation only: the search allocates what is missing
			g_timeline.mark("seed stage of block " + std::to_string(bi) + " starts");
			t0 = std::chrono::steady_clock::now();
			int64_t n_hits = 0;
			chk(dmnd_seed_search(ctx, &sp, &n_hits));
			g_timeline.mark("dmnd_seed_search returned");
			std::vector<dmnd_seed_hit> hits((size_t)n_hits);
			chk(dmnd_seed_hits(ctx, hits.data(), n_hits));
			const double sd = ms_since(t0);
			if (g_timeline.on) {
				double ms[5] = { 0, 0, 0, 0, 0 };
				(void)dmnd_seed_kernel_ms(ctx, ms);
				char b[160];
				std::snprintf(b, sizeof b, " (kernels: index %.2f, stream %.2f, mask %.2f, pairs %.2f, all %.2f ms)", ms[0], ms[1], ms[2], ms[3], ms[4]);
				g_timeline.mark("seed stage of block " + std::to_string(bi) + " done" + b);
			}
			if (o.global_ranking > 0) {
				// no extension here: the block pair's seed hits only update the ranking table (search/stage2.h:138, table.cpp:153-190);
				// the targets are neither masked lazily (extend.cpp:202) nor extended before the last block
				std::vector<dmnd_ranked_target> recs((size_t)std::max<int64_t>(n_hits, 1));
				int64_t n_recs = 0;
				chk(dmnd_rank_targets(ctx, q.data.data(), t.data.data(), hits.data(), n_hits, threads, recs.data(), (int64_t)recs.size(), &n_recs));
				for (int64_t k = 0; k < n_recs; ++k) recs[(size_t)k].target += (uint32_t)tr.begin;
				std::lock_guard<std::mutex> lock(merge_mutex);
				chk(dmnd_rank_update(rank_table.data(), (int64_t)(qr.end - qr.begin), o.global_ranking, recs.data(), n_recs));
				for (const dmnd_seed_hit& h : hits) seeded[h.query / C] = 1;
				ms_upload += up; ms_mask += mk; ms_seed += sd;
				total_hits += n_hits;
				if (&qr == &q_blocks.front()) { mt_total += mt; motif_letters += ml; }
				g_timeline.mark("ranking table updated with block " + std::to_string(bi));
				continue;
			}
			if (lazy_masking) {
				if (!seg) {                                      // which targets: the sequence that holds each hit's reference position
					// (a binary search over 10^6 limits per hit: 8 ms for 84 000 hits on one thread -- a team marks the targets, the marks
					// are collected in order)
					std::vector<uint8_t> seen(t.limits.size(), 0);
					const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, hits.size() / 4096));
					std::vector<std::thread> team;
					for (int w = 0; w < T; ++w)
						team.emplace_back([&, w] {
							for (size_t k = hits.size() * (size_t)w / (size_t)T, e = hits.size() * (size_t)(w + 1) / (size_t)T; k < e; ++k)
								seen[(size_t)(std::upper_bound(t.limits.begin(), t.limits.end(), hits[k].subject) - t.limits.begin() - 1)] = 1;
						});
					for (std::thread& th : team) th.join();
					for (size_t i = 0; i + 1 < t.limits.size(); ++i) if (seen[i]) lazy_ids.push_back((int32_t)i);
				}
				mask_target();
				g_timeline.mark("reference block " + std::to_string(bi) + " masked lazily (" + std::to_string(lazy_ids.size()) + " targets)");
			}
			t0 = std::chrono::steady_clock::now();
			if (o.no_self_hits) {
				if (blastx) throw std::runtime_error("--no-self-hits is not supported for blastx");      // basic/config.cpp:677
				self_ctx[(size_t)g] = SelfCtx{ &qtitles, &db, qr.begin, tr.begin };
				chk(dmnd_set_no_self_hits(ctx, [](void* u, uint32_t q, uint32_t t) -> int {
					const SelfCtx& s = *static_cast<const SelfCtx*>(u);
					return (*s.qtitles)[s.q0 + q] == s.db->title(s.ordinals ? (size_t)(*s.ordinals)[t] : s.t0 + t) ? 1 : 0;
				}, &self_ctx[(size_t)g]));
			}
			std::vector<dmnd_match> mine((size_t)std::max<int64_t>(n_hits, 1));
			std::vector<uint8_t> my_arena;
			int64_t n_matches = 0;
			// (with --max-hsps a target can have more HSP records than seed hits: the call reports the number it needs)
			if (!need_transcripts) {
				int rc = dmnd_extend(ctx, q.data.data(), t_host, hits.data(), n_hits, threads, 0, mine.data(), (int64_t)mine.size(), &n_matches, nullptr, 0, nullptr);
				if (rc == DMND_E_CAP && n_matches > (int64_t)mine.size()) {
					mine.resize((size_t)n_matches);
					rc = dmnd_extend(ctx, q.data.data(), t_host, hits.data(), n_hits, threads, 0, mine.data(), (int64_t)mine.size(), &n_matches, nullptr, 0, nullptr);
				}
				chk(rc);
			}
			else {
				int64_t cap = std::max<int64_t>((int64_t)1 << 20, 64 * n_hits), used = 0;
				for (;;) {
					my_arena.resize((size_t)cap);
					const int rc = dmnd_extend(ctx, q.data.data(), t_host, hits.data(), n_hits, threads, 0, mine.data(), (int64_t)mine.size(), &n_matches, my_arena.data(), cap, &used);
					if (rc == DMND_E_CAP && n_matches > (int64_t)mine.size()) { mine.resize((size_t)n_matches); continue; }
					if (rc == DMND_E_CAP && cap < ((int64_t)1 << 36)) { cap *= 4; continue; }
					chk(rc);
					break;
				}
				my_arena.resize((size_t)used);
			}
			mine.resize((size_t)n_matches);
			const double ex = ms_since(t0);
			g_timeline.mark("extension of block " + std::to_string(bi) + " done");
			// into the query block's record list: block ids -> file ordinals, transcripts appended to the block's arena
			std::lock_guard<std::mutex> lock(merge_mutex);
			for (const dmnd_seed_hit& h : hits) seeded[h.query / C] = 1;
			for (dmnd_match& m : mine) {
				m.query += (uint32_t)qr.begin; m.target += (uint32_t)tr.begin;
				if (need_transcripts) m.hsp.transcript_off += (int64_t)arena.size();
			}
			joined.insert(joined.end(), mine.begin(), mine.end());
			joined_gpu.insert(joined_gpu.end(), mine.size(), (uint8_t)g);
			arena.insert(arena.end(), my_arena.begin(), my_arena.end());
			ms_upload += up; ms_mask += mk; ms_seed += sd; ms_ext += ex;
			total_hits += n_hits;
			if (&qr == &q_blocks.front()) { mt_total += mt; motif_letters += ml; }
		}
		});
		if (o.global_ranking > 0) {
			// Extension::GlobalRanking::extend (global_ranking/extend.cpp:171-233): the ranked targets of this query block as ONE reference
			// block (database order), masked as the option says, then Extension::extend per query over the table's entries -- one seed
			// hit per entry carrying its score and context -- full matrix, no ranking chunks, no gapped filter
			auto t0 = std::chrono::steady_clock::now();
			std::vector<uint32_t> ordinals;
			for (const dmnd_ranked_target& e : rank_table) if (e.score) ordinals.push_back(e.target);
			std::sort(ordinals.begin(), ordinals.end());
			ordinals.erase(std::unique(ordinals.begin(), ordinals.end()), ordinals.end());
			std::cerr << "#Ranked database sequences: " << ordinals.size() << "\n";
			if (!ordinals.empty()) {
				dmnd_ctx* ctx = ctxs[0];
				SeqBlock rb;
				rb.begin();
				for (uint32_t oid : ordinals) rb.push(db.sequence(oid), std::string());
				rb.finish();
				int64_t mt = 0;
				if (seg) chk(dmnd_seg_mask_block(rb.data.data(), rb.limits.data(), (int64_t)ordinals.size(), threads, &mt));
				chk(dmnd_upload_block(ctx, DMND_TARGET, rb.data.data(), (int64_t)rb.data.size(), rb.limits.data(), (int64_t)ordinals.size()));
				if (tantan) chk(dmnd_mask_block(ctx, DMND_TARGET, rb.data.data(), &mt));
				held_blocks[0].index = (size_t)-1; held_blocks[0].ahead = false;      // the reference block this GPU held is no longer in HBM
				std::vector<dmnd_seed_hit> hits;
				for (size_t qi = 0; qi < qr.end - qr.begin; ++qi)
					for (int k = 0; k < o.global_ranking; ++k) {
						const dmnd_ranked_target& e = rank_table[qi * (size_t)o.global_ranking + (size_t)k];
						if (!e.score) break;
						const size_t local = (size_t)(std::lower_bound(ordinals.begin(), ordinals.end(), e.target) - ordinals.begin());
						dmnd_seed_hit h;
						std::memset(&h, 0, sizeof h);
						h.query = (uint32_t)(qi * C + e.context); h.seed_offset = 0; h.subject = rb.limits[local]; h.score = e.score;
						hits.push_back(h);
					}
				const int64_t n_hits = (int64_t)hits.size();
				if (o.no_self_hits) {
					if (blastx) throw std::runtime_error("--no-self-hits is not supported for blastx");
					self_ctx[0] = SelfCtx{ &qtitles, &db, qr.begin, 0, &ordinals };
					chk(dmnd_set_no_self_hits(ctx, [](void* u, uint32_t q, uint32_t t) -> int {
						const SelfCtx& s = *static_cast<const SelfCtx*>(u);
						return (*s.qtitles)[s.q0 + q] == s.db->title((size_t)(*s.ordinals)[t]) ? 1 : 0;
					}, &self_ctx[0]));
				}
				joined.assign((size_t)std::max<int64_t>(n_hits, 1), dmnd_match());
				int64_t n_m = 0, cap = std::max<int64_t>((int64_t)1 << 20, 64 * n_hits), used = 0;
				for (;;) {
					if (need_transcripts) arena.resize((size_t)cap);
					const int rc = dmnd_extend(ctx, q.data.data(), rb.data.data(), hits.data(), n_hits, threads, 0, joined.data(), (int64_t)joined.size(), &n_m,
						need_transcripts ? arena.data() : nullptr, need_transcripts ? cap : 0, need_transcripts ? &used : nullptr);
					if (rc == DMND_E_CAP && n_m > (int64_t)joined.size()) { joined.resize((size_t)n_m); continue; }
					if (rc == DMND_E_CAP && need_transcripts && cap < ((int64_t)1 << 36)) { cap *= 4; continue; }
					chk(rc);
					break;
				}
				joined.resize((size_t)n_m);
				arena.resize(need_transcripts ? (size_t)used : 0);
				for (dmnd_match& m : joined) { m.query += (uint32_t)qr.begin; m.target = ordinals[m.target]; }
				mt_total += mt;
			}
			ms_ext += ms_since(t0);
			g_timeline.mark("globally ranked targets extended");
		}
		g_timeline.mark("block pairs of the query block done");
		int64_t n_matches = (int64_t)joined.size();
		// Several GPUs: the final merge runs over RCCL and on the devices (dmnd_join_ranks: the records of every GPU go to the owner of
		// their query range in one grouped ncclSend / ncclRecv exchange and are merged there); one record per (query, target) only.
		// DMND_CLI_RCCL=1 takes that path with one GPU too (RCCL with itself), =0 keeps the host join.
		static const int rccl_env = [] { const char* e = std::getenv("DMND_CLI_RCCL"); return e ? std::atoi(e) : -1; }();
		const bool rank_join = t_blocks.size() > 1 && o.global_ranking == 0 && o.max_hsps == 1 && !o.range_culling && !joined.empty()
			&& (rccl_env >= 0 ? rccl_env != 0 : n_gpus > 1);
		bool rank_joined = false;
		if (rank_join) {
			std::vector<std::vector<dmnd_match>> per_gpu((size_t)n_gpus);
			for (size_t k = 0; k < joined.size(); ++k) {
				dmnd_match m = joined[k];
				m.query -= (uint32_t)qr.begin;                      // owners by the query block's own range
				per_gpu[joined_gpu[k]].push_back(m);
			}
			std::vector<const dmnd_match*> ptrs((size_t)n_gpus);
			std::vector<int64_t> counts((size_t)n_gpus);
			for (int g = 0; g < n_gpus; ++g) { ptrs[(size_t)g] = per_gpu[(size_t)g].data(); counts[(size_t)g] = (int64_t)per_gpu[(size_t)g].size(); }
			int transport = 0;
			std::vector<dmnd_match> merged(joined.size());
			const int jrc = dmnd_join_ranks(ctxs.data(), n_gpus, ptrs.data(), counts.data(), (int64_t)(qr.end - qr.begin), o.k, o.top, merged.data(), (int64_t)merged.size(), &n_matches, &transport);
			if (jrc == DMND_OK) {
				for (int64_t k = 0; k < n_matches; ++k) { joined[(size_t)k] = merged[(size_t)k]; joined[(size_t)k].query += (uint32_t)qr.begin; }
				if (&qr == &q_blocks.front()) std::cerr << "Block join: transport_used = " << (transport == 1 ? "RCCL exchange (ncclSend/ncclRecv) between " : "device-to-device copies between ") << n_gpus
					<< " context(s), merged on the device(s)\n";
				rank_joined = true;
			}
			else if (jrc == DMND_E_DEVICE || jrc == DMND_E_CAP || jrc == DMND_E_NOMEM) {
				// no RCCL library, no peer access, ncclCommInitAll refused by the container ...: the host join gives the same records
				std::cerr << "Block join: the device exchange is not available (" << dmnd_last_error() << "); transport_used = host join\n";
				n_matches = (int64_t)joined.size();
			}
			else chk(jrc);
		}
		if (!rank_joined)
		// the join's culler is the one TargetCulling::get picks (output/target_culling.cpp:22-28): RangeCulling with --range-culling
		if (t_blocks.size() > 1 && o.global_ranking == 0) chk(o.range_culling ? dmnd_join_blocks_range(joined.data(), (int64_t)joined.size(), o.k, o.top, o.range_cover, &n_matches)
			: o.top >= 0.0 ? dmnd_join_blocks_top(joined.data(), (int64_t)joined.size(), o.top, &n_matches)
			: dmnd_join_blocks(joined.data(), (int64_t)joined.size(), o.k, &n_matches));
		if (t_blocks.size() > 1 && o.frameshift > 0 && need_transcripts) {
			// Several reference blocks: the reference prints from the joined IntermediateRecords, and with a frameshift penalty those
			// carry transcripts (output_format.cpp:256-257), so every printed Hsp is re-counted from its transcript by HspContext::parse
			// (basic/hssp.cpp:64-100) -- where a frameshift operation is a column of the alignment: `length` (and pident with it) counts
			// it, which the traceback's own statistics of a single-block run do not. E-value and bit score stay the record's.
			for (int64_t k = 0; k < n_matches; ++k) {
				dmnd_match& m = joined[(size_t)k];
				const size_t base = (size_t)m.query * C + (size_t)(m.frame / 3) * 3 - qr.begin * C;
				const int8_t* f3[3];
				int32_t l3[3];
				for (int j = 0; j < 3; ++j) { f3[j] = q.data.data() + q.limits[base + (size_t)j]; l3[j] = (int32_t)(q.limits[base + (size_t)j + 1] - q.limits[base + (size_t)j] - 1); }
				const double ev = m.evalue, bs = m.bit_score;
				chk(dmnd_hsp_from_transcript_frames(&p, f3, l3, source_len[m.query], l3[0], (int32_t)db.length(m.target), arena.data() + m.hsp.transcript_off, &m));
				m.evalue = ev; m.bit_score = bs;
			}
		}
		std::vector<int8_t> full_sseq_buf;                    // the unmasked target of the line being printed (full_sseq)
		auto view_of = [&](const dmnd_match& m) {
			dmnd_hsp_view v;
			const size_t ctx_id = (size_t)m.query * C + (size_t)m.frame;            // the aligned query context in the file
			v.match = &m;
			v.transcript = need_transcripts ? arena.data() + m.hsp.transcript_off : nullptr;
			v.qtitle = qtitles[m.query].c_str(); v.stitle = db.title(m.target).c_str();
			// the query block at hand holds the letters the extension stage saw (tantan-masked in place)
			const size_t local = ctx_id - qr.begin * C;
			v.qseq = q.data.data() + q.limits[local]; v.qlen = (int32_t)(q.limits[local + 1] - q.limits[local] - 1);
			v.slen = (int32_t)db.length(m.target);
			if (want_full_sseq) { full_sseq_buf = db.sequence(m.target); v.full_sseq = full_sseq_buf.data(); } else v.full_sseq = nullptr;
			v.source_seq = blastx ? reads[m.query].data() : nullptr; v.source_len = blastx ? source_len[m.query] : 0;
			v.qnum = (int64_t)m.query; v.snum = (int64_t)m.target;
			// the three reading frames of the alignment's strand: a frameshift alignment walks through them (contexts 0-2 / 3-5 of the read)
			for (int k = 0; k < 3; ++k) {
				const size_t c3 = (size_t)m.query * C + (size_t)(m.frame / 3) * 3 + (size_t)k - qr.begin * C;
				v.qframes[k] = blastx && o.frameshift > 0 ? q.data.data() + q.limits[c3] : nullptr;
			}
			return v;
		};
		std::vector<char> big;
		auto put = [&](int64_t w, const char* p) { if (w < 0) throw std::runtime_error(dmnd_last_error()); out.write(p, (size_t)w); };
		int64_t i = 0;
		// The pairwise and PAF formats also report queries without alignments, in query order (DEFAULT_REPORT_UNALIGNED): with one
		// reference block only those that had seed hits (a query without any is skipped before the output stage, align/align.cpp:173-176,
		// align/output.cpp:35-53), with several blocks every one (output/join_blocks.cpp:302-308,365-372).
		const bool per_query = fmt == FMT_PAIRWISE || fmt == FMT_PAF || fmt == FMT_SAM || fmt == FMT_XML || fmt == FMT_DAA || (fmt == FMT_FIELDS && report_unal);
		for (size_t qi = qr.begin; qi < qr.end && per_query; ++qi) {
			const bool has = i < n_matches && joined[(size_t)i].query == (uint32_t)qi;
			// (with a frameshift penalty the legacy pipeline is entered for every query, a query without seed hits gets its intro there:
			// align/align.cpp:167-171,119-129)
			if (!has && (!report_unal || (t_blocks.size() == 1 && !seeded[qi - qr.begin] && o.frameshift == 0))) continue;
			const int32_t qlen = blastx ? source_len[qi] : (int32_t)(q_all.limits[qi + 1] - q_all.limits[qi] - 1);
			big.resize(qtitles[qi].size() + 256);
			if (fmt == FMT_PAIRWISE) put(dmnd_format_pairwise_intro(qtitles[qi].c_str(), qlen, has ? 0 : 1, big.data(), (int64_t)big.size()), big.data());
			else if (fmt == FMT_XML) { big.resize(qtitles[qi].size() * 6 + 512); put(dmnd_format_xml_query_intro(qtitles[qi].c_str(), (int64_t)qi, qlen, big.data(), (int64_t)big.size()), big.data()); }
			else if (!has && fmt == FMT_FIELDS) {
				const size_t local = (qi - qr.begin) * C;
				const int32_t l0 = (int32_t)(q.limits[local + 1] - q.limits[local] - 1);
				big.resize(qtitles[qi].size() * 2 + (size_t)(blastx ? source_len[qi] : l0) + 4096);
				put(dmnd_format_fields_unaligned(qtitles[qi].c_str(), q.data.data() + q.limits[local], l0, blastx ? reads[qi].data() : nullptr, blastx ? source_len[qi] : 0,
					field_ids.data(), (int)field_ids.size(), big.data(), (int64_t)big.size()), big.data());
			}
			else if (!has) put(fmt == FMT_SAM ? dmnd_format_sam(nullptr, qtitles[qi].c_str(), big.data(), (int64_t)big.size())
				: dmnd_format_paf(nullptr, qtitles[qi].c_str(), big.data(), (int64_t)big.size()), big.data());
			int32_t xml_hit = -1, xml_hsp = 0;                   // records of one target follow each other (--max-hsps): Hit_num / Hsp_num
			uint32_t xml_target = UINT32_MAX;
			std::string daa_rec;
			if (fmt == FMT_DAA) {
				const size_t local = (qi - qr.begin) * C;
				const int8_t* letters = blastx ? reads[qi].data() : q.data.data() + q.limits[local];
				const int32_t n = blastx ? source_len[qi] : (int32_t)(q.limits[local + 1] - q.limits[local] - 1);
				big.resize(qtitles[qi].size() + (size_t)n + 64);
				const int64_t w = dmnd_format_daa_query(qtitles[qi].c_str(), letters, n, blastx ? 1 : 0, big.data(), (int64_t)big.size());
				if (w < 0) throw std::runtime_error(dmnd_last_error());
				daa_rec.assign(big.data(), (size_t)w);
			}
			for (; i < n_matches && joined[(size_t)i].query == (uint32_t)qi; ++i) {
				const dmnd_match& m = joined[(size_t)i];
				const dmnd_hsp_view v = view_of(m);
				big.resize((size_t)m.hsp.length * 8 + std::strlen(v.qtitle) + std::strlen(v.stitle) + 4096);
				if (fmt == FMT_PAIRWISE) put(dmnd_format_pairwise(&v, p.matrix8, big.data(), (int64_t)big.size()), big.data());
				else if (fmt == FMT_DAA) {
					if (daa_dict_id[m.target] == UINT32_MAX) { daa_dict_id[m.target] = (uint32_t)daa_dict.size(); daa_dict.push_back(m.target); }
					big.resize((size_t)m.hsp.transcript_len + 64);
					const int64_t w = dmnd_format_daa_match(&v, daa_dict_id[m.target], big.data(), (int64_t)big.size());
					if (w < 0) throw std::runtime_error(dmnd_last_error());
					daa_rec.append(big.data(), (size_t)w);
				}
				else if (fmt == FMT_XML) {
					big.resize((size_t)m.hsp.length * 4 + 6 * std::strlen(v.stitle) + 4096);
					if (m.target == xml_target) ++xml_hsp; else { ++xml_hit; xml_hsp = 0; xml_target = m.target; }
					put(dmnd_format_xml(&v, xml_hit, xml_hsp, p.matrix8, big.data(), (int64_t)big.size()), big.data());
				}
				else if (fmt == FMT_FIELDS) {
					big.resize((size_t)m.hsp.length * 4 + (size_t)v.qlen * 3 + (size_t)v.slen + std::strlen(v.qtitle) + 2 * std::strlen(v.stitle) + (size_t)v.source_len * 2 + 4096);
					put(dmnd_format_fields(&v, field_ids.data(), (int)field_ids.size(), big.data(), (int64_t)big.size()), big.data());
				}
				else if (fmt == FMT_SAM) put(dmnd_format_sam(&v, nullptr, big.data(), (int64_t)big.size()), big.data());
				else put(dmnd_format_paf(&v, nullptr, big.data(), (int64_t)big.size()), big.data());
			}
			if (fmt == FMT_DAA) {                                               // finish_daa_query_record: byte count after the size field
				const uint32_t size = (uint32_t)(daa_rec.size() - 4);
				std::memcpy(&daa_rec[0], &size, 4);
				out.write(daa_rec);
				daa_bytes += (int64_t)daa_rec.size();
				++daa_queries;
			}
			if (fmt == FMT_XML) put(dmnd_format_xml_query_epilog(has ? 0 : 1, (int64_t)db.n, db.letters, p.K, p.lambda, big.data(), (int64_t)big.size()), big.data());
			if (has) ++aligned;
		}
		// one line per record, no state between records: the records are cut into as many runs as there are host threads, every
		// thread formats its run into a string, the strings go to the file in order (12 000 lines: 19 ms on one thread)
		if (!per_query && !want_full_sseq && n_matches >= 2048 && threads > 1) {
			const int T = (int)std::min<int64_t>(threads, n_matches / 1024);
			std::vector<std::string> text((size_t)T);
			std::vector<std::string> errors((size_t)T);
			std::vector<std::thread> team;
			for (int t = 0; t < T; ++t)
				team.emplace_back([&, t] {
					const int64_t b = n_matches * t / T, e = n_matches * (t + 1) / T;
					std::string& s = text[(size_t)t];
					s.reserve((size_t)(e - b) * 96);
					std::vector<char> buf;
					char one[8192];
					for (int64_t k = b; k < e; ++k) {
						const dmnd_match& m = joined[(size_t)k];
						int64_t w;
						const char* from;
						if (fmt == FMT_FIELDS) {
							const dmnd_hsp_view v = view_of(m);
							const size_t need = (size_t)m.hsp.length * 4 + (size_t)v.qlen * 3 + (size_t)v.slen + std::strlen(v.qtitle) + 2 * std::strlen(v.stitle) + (size_t)v.source_len * 2 + 4096;
							if (buf.size() < need) buf.resize(need);
							w = dmnd_format_fields(&v, field_ids.data(), (int)field_ids.size(), buf.data(), (int64_t)buf.size());
							from = buf.data();
						}
						else {
							w = blastx ? dmnd_format_tab_translated(&m, qid[m.query].c_str(), short_id(db.title(m.target)).c_str(), source_len[m.query], one, sizeof one)
								: dmnd_format_tab(&m, qid[m.query].c_str(), short_id(db.title(m.target)).c_str(), one, sizeof one);
							from = one;
						}
						if (w < 0) { const char* why = dmnd_last_error(); errors[(size_t)t] = why && why[0] ? why : "formatting a match record failed"; return; }      // (never empty: the main thread tests for it)
						s.append(from, (size_t)w);
					}
				});
			for (std::thread& th : team) th.join();
			g_timeline.mark("records formatted by " + std::to_string(T) + " threads");
			for (int t = 0; t < T; ++t) {
				if (!errors[(size_t)t].empty()) throw std::runtime_error(errors[(size_t)t]);
				out.write(text[(size_t)t].data(), text[(size_t)t].size());
			}
			for (int64_t k = 0; k < n_matches; ++k)
				if (k == 0 || joined[(size_t)k].query != joined[(size_t)k - 1].query) ++aligned;
			i = n_matches;
		}
		for (; i < n_matches && !per_query; ++i) {
			const dmnd_match& m = joined[(size_t)i];
			if (fmt == FMT_FIELDS) {
				const dmnd_hsp_view v = view_of(m);
				big.resize((size_t)m.hsp.length * 4 + (size_t)v.qlen * 3 + (size_t)v.slen + std::strlen(v.qtitle) + 2 * std::strlen(v.stitle) + (size_t)v.source_len * 2 + 4096);
				put(dmnd_format_fields(&v, field_ids.data(), (int)field_ids.size(), big.data(), (int64_t)big.size()), big.data());
			}
			else {
				const int w = blastx ? dmnd_format_tab_translated(&m, qid[m.query].c_str(), short_id(db.title(m.target)).c_str(), source_len[m.query], line, sizeof line)
					: dmnd_format_tab(&m, qid[m.query].c_str(), short_id(db.title(m.target)).c_str(), line, sizeof line);
				put(w, line);
			}
			if (i == 0 || joined[(size_t)i].query != joined[(size_t)i - 1].query) ++aligned;
		}
		total_matches += n_matches;
		// --un / --al: the block's queries without / with a reported alignment, in file order, as FASTA wrapped at 160 letters with
		// the letters of the block as it stands after masking (write_unaligned / write_aligned, data/queries.cpp:32-66; a read of a
		// translated search as its DNA)
		if (un_file || al_file) {
			std::vector<uint8_t> has((size_t)(qr.end - qr.begin), 0);
			for (int64_t k = 0; k < n_matches; ++k) has[(size_t)joined[(size_t)k].query - qr.begin] = 1;
			std::string rec;
			for (size_t qi = qr.begin; qi < qr.end; ++qi) {
				FILE* f = has[qi - qr.begin] ? al_file : un_file;
				if (!f) continue;
				rec.assign(1, '>'); rec += qtitles[qi]; rec += '\n';
				const size_t local = (qi - qr.begin) * C;
				const int8_t* letters = blastx ? reads[qi].data() : q.data.data() + q.limits[local];
				const int64_t n = blastx ? (int64_t)source_len[qi] : q.limits[local + 1] - q.limits[local] - 1;
				const char* alphabet = blastx ? "ACGTN" : "ARNDCQEGHILKMFPSTWYVBJZX*_";
				for (int64_t x = 0; x < n; x += 160) {
					for (int64_t y = x; y < std::min(x + 160, n); ++y) rec += alphabet[letters[y] & 31];
					rec += '\n';
				}
				if (std::fwrite(rec.data(), 1, rec.size(), f) != rec.size()) throw std::runtime_error("Error writing file");
			}
		}
	}
	if (fmt == FMT_XML) out.write("</BlastOutput_iterations>\n</BlastOutput>");        // XMLFormat::print_footer
	if (fmt == FMT_DAA) {                                                              // finish_daa, daa_write.cpp:75-123
		const uint32_t zero = 0;
		out.write((const char*)&zero, 4);
		daa_bytes += 4;
		int64_t name_bytes = 0;
		for (uint32_t t : daa_dict) {
			const std::string& title = db.title(t);
			std::string name;
			if (o.salltitles) name = title;
			else if (o.sallseqid) {                                                     // Util::Seq::all_seqids: the id of every title of the record
				size_t b = 0;
				for (;;) {
					const size_t e = title.find('\1', b);
					if (b > 0) name += '\1';
					name += short_id(title.substr(b, e == std::string::npos ? std::string::npos : e - b));
					if (e == std::string::npos) break;
					b = e + 1;
				}
			}
			else name = title.substr(0, std::strcspn(title.c_str(), " \a\b\f\n\r\t\v\1"));
			out.write(name.c_str(), name.size() + 1);
			name_bytes += (int64_t)name.size() + 1;
		}
		for (uint32_t t : daa_dict) { const uint32_t len = (uint32_t)db.length(t); out.write((const char*)&len, 4); }
		daa.db_seqs_used = (int64_t)daa_dict.size(); daa.query_records = daa_queries; daa.finished = 1; daa.alignment_bytes = daa_bytes; daa.ref_name_bytes = name_bytes;
		std::vector<char> hb(4096);
		const int64_t w = dmnd_format_daa_header(&daa, hb.data(), (int64_t)hb.size());
		if (w < 0) throw std::runtime_error(dmnd_last_error());
		if (std::fseek(out.f, 0, SEEK_SET) != 0) throw std::runtime_error("Error writing the DAA header");
		out.write(hb.data(), (size_t)w);
	}
	g_timeline.mark("records formatted");
	out.close();
	if (un_file) std::fclose(un_file);
	if (al_file) std::fclose(al_file);
	g_timeline.mark("output written");
	// The results are on disk. Tearing the contexts down buffer by buffer and unloading the HIP runtime costs tens of
	// milliseconds that buy nothing -- the driver reclaims a process's HBM when it exits -- so the process leaves through _exit
	// after the report below (DMND_CLI_CLEAN_EXIT=1: full teardown, for leak checkers).
	// A process that a profiler, tracer, sanitizer or coverage tool looks into leaves the normal way on its own: those do their
	// end-of-run work in exit handlers (a rocprofv3 run of this binary that left through _exit never finished).
	bool clean_exit = std::getenv("DMND_CLI_CLEAN_EXIT") != nullptr;
	for (const char* v : { "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "ROCPROF_OUTPUT_PATH", "ROCPROFILER_METRICS_PATH", "HSA_TOOLS_LIB", "ROCTRACER_DOMAIN", "ASAN_OPTIONS", "TSAN_OPTIONS", "GCOV_PREFIX", "LLVM_PROFILE_FILE" })
		if (std::getenv(v)) clean_exit = true;
	if (const char* pre = std::getenv("LD_PRELOAD")) if (std::strstr(pre, "rocprof") || std::strstr(pre, "roctracer") || std::strstr(pre, "asan")) clean_exit = true;
	if (clean_exit) for (dmnd_ctx* c : ctxs) dmnd_destroy(c);
	std::cerr << "Uploading blocks to HBM...  [" << ms_upload / 1e3 << "s]\n";
	if (motifs) std::cerr << "Soft-masked letters (motifs): " << motif_letters << "\n";
	if (seg) std::cerr << "Masking reference (seg)...  [" << ms_mask / 1e3 << "s]  masked letters: " << mt_total << "\n";
	if (tantan) std::cerr << "Masking queries and reference (tantan)...  [" << ms_mask / 1e3 << "s]  masked letters: " << mq_total << " + " << mt_total << "\n";
	std::cerr << "Searching alignments (seed stage)...  [" << ms_seed / 1e3 << "s]  hits=" << total_hits << "\n";
	std::cerr << "Computing alignments (extension stage)...  [" << ms_ext / 1e3 << "s]\n";
	std::cerr << "Total time = " << ms_since(t_all) / 1e3 << "s\nReported " << total_matches << " pairwise alignments, " << total_matches << " HSPs.\n" << aligned << " queries aligned.\n";
	g_timeline.print();
	if (!clean_exit) { std::cerr.flush(); std::cout.flush(); std::fflush(nullptr); ::_exit(0); }
	return 0;
}

// `view`: prints a DAA archive in another format (view_daa / view_query, src/legacy/daa/view.cpp:88-186). Host only: the records hold
// score, begin coordinates and transcript; everything else is recomputed from them (dmnd_hsp_from_transcript).
int run_view(const Options& o)
{
	if (o.daa.empty()) throw std::runtime_error("The view command requires a DAA (option -a) input file.");
	std::string path = o.daa;
	if (!std::ifstream(path).good() && std::ifstream(path + ".daa").good()) path += ".daa";
	std::ifstream f(path, std::ios::binary | std::ios::ate);
	if (!f) throw std::runtime_error("Error opening file " + path);
	std::vector<uint8_t> file((size_t)f.tellg());
	f.seekg(0);
	if (!file.empty() && !f.read((char*)file.data(), (std::streamsize)file.size())) throw std::runtime_error("Error reading file " + path);
	const size_t H1 = 16, H2 = 2432, HEAD = H1 + H2;
	auto rd64 = [&](size_t off) { uint64_t x; std::memcpy(&x, file.data() + off, 8); return x; };
	auto rd32 = [&](size_t off) { int32_t x; std::memcpy(&x, file.data() + off, 4); return x; };
	auto rdd = [&](size_t off) { double x; std::memcpy(&x, file.data() + off, 8); return x; };
	if (file.size() < HEAD || rd64(0) != 0x3c0e53476d3ee36bULL) throw std::runtime_error("Input file is not a DAA file.");
	if (rd64(8) > 1) throw std::runtime_error("DAA version requires later version of DIAMOND.");
	// DAA_header2 (legacy/daa/daa_file.h:41-90): build, db_seqs, db_seqs_used, db_letters, flags, query_records; mode, gap_open, gap_extend, ...
	const uint64_t db_seqs = rd64(H1 + 8), used = rd64(H1 + 16), db_letters = rd64(H1 + 24);
	const int32_t mode = rd32(H1 + 48), gap_open = rd32(H1 + 52), gap_extend = rd32(H1 + 56);
	const double max_evalue = rdd(H1 + 96);
	char matrix[17] = { 0 };
	std::memcpy(matrix, file.data() + H1 + 112, 16);
	const uint64_t aln_bytes = rd64(H1 + 128), name_bytes = rd64(H1 + 136);
	if (aln_bytes == 0) throw std::runtime_error("Invalid DAA file. DIAMOND run has probably not completed successfully.");
	if (mode != 2 && mode != 3) throw std::runtime_error("This build reads blastp and blastx archives.");
	const bool blastx = mode == 3;
	if (HEAD + aln_bytes + name_bytes + used * 4 > file.size()) throw std::runtime_error("Truncated DAA file.");
	std::vector<const char*> ref_name((size_t)used);
	{
		const char* p = (const char*)file.data() + HEAD + aln_bytes;
		const char* const end = p + name_bytes;
		for (uint64_t i = 0; i < used; ++i) {
			if (p >= end) throw std::runtime_error("Truncated DAA file.");
			ref_name[(size_t)i] = p;
			p += std::strlen(p) + 1;
		}
	}
	const uint8_t* ref_len = file.data() + HEAD + aln_bytes + name_bytes;
	dmnd_params p;
	dmnd_default_params(&p);
	if (dmnd_matrix_params(matrix, gap_open, gap_extend, &p) != DMND_OK) throw std::runtime_error(dmnd_last_error());
	p.db_letters = (double)db_letters;
	p.max_evalue = max_evalue;
	std::cerr << "Scoring parameters: (Matrix=" << matrix << " Lambda=" << p.lambda << " K=" << p.K << " Penalties=" << gap_open << "/" << gap_extend << ")\nDB sequences = " << db_seqs
		<< "\nDB sequences used = " << used << "\nDB letters = " << db_letters << "\n";

	enum { V_FIELDS, V_PAIRWISE, V_XML, V_SAM, V_PAF } fmt = V_FIELDS;
	std::vector<int32_t> field_ids;
	int need_tr = 0;
	std::vector<const char*> names;
	const std::string f0 = o.outfmt.empty() ? "6" : o.outfmt[0];
	if (f0 == "6" || f0 == "tab") {
		static const char* const std_fields[12] = { "qseqid", "sseqid", "pident", "length", "mismatch", "gapopen", "qstart", "qend", "sstart", "send", "evalue", "bitscore" };
		if (o.outfmt.size() > 1) for (size_t i = 1; i < o.outfmt.size(); ++i) names.push_back(o.outfmt[i].c_str());
		else names.assign(std_fields, std_fields + 12);
		field_ids.resize(names.size());
		if (dmnd_output_fields(names.data(), (int)names.size(), field_ids.data(), &need_tr) != DMND_OK) throw std::runtime_error(dmnd_last_error());
		for (int32_t id : field_ids)
			if (id == DMND_F_FULL_SSEQ) throw std::runtime_error("full_sseq is not stored in a DAA file");
	}
	else if (f0 == "0" || f0 == "pairwise") fmt = V_PAIRWISE;
	else if (f0 == "5" || f0 == "xml") fmt = V_XML;
	else if (f0 == "101" || f0 == "sam") fmt = V_SAM;
	else if (f0 == "103" || f0 == "paf") fmt = V_PAF;
	else throw std::runtime_error("Invalid output format: " + f0 + " (view prints 6, 0, 5, 101 and 103)");
	if (fmt != V_FIELDS && o.outfmt.size() > 1) throw std::runtime_error("Invalid output format: only the tabular format takes fields");

	Sink out;
	{
		std::string op = o.out;
		if (o.compress && (op.size() < 3 || op.substr(op.size() - 3) != ".gz")) op += ".gz";
		out.open(op, o.compress);
	}
	std::vector<char> big(1 << 16);
	auto put = [&](int64_t w) { if (w < 0) throw std::runtime_error(dmnd_last_error()); out.write(big.data(), (size_t)w); };
	if (fmt == V_PAIRWISE) out.write("BLASTP 2.3.0+\n\n\n");
	if (fmt == V_SAM) {
		const std::string prog = blastx ? "BlastX" : "BlastP";
		out.write("@HD\tVN:1.5\tSO:query\n@PG\tPN:diamond-hip\tVN:ABI" + std::to_string(dmnd_abi_version()) + "\n@mm\t" + prog + "\n@CO\t" + prog
			+ "-like alignments\n@CO\tReporting AS: bitScore, ZR: rawScore, ZE: expected, ZI: percent identity, ZL: reference length, ZF: frame, ZS: query start DNA coordinate\n");
	}
	const int64_t K = o.k > 0 ? o.k : 25;
	size_t pos = HEAD;
	const size_t aln_end = HEAD + (size_t)aln_bytes;
	int64_t qnum = 0, n_hsps = 0;
	const bool check_daa = std::getenv("DMND_VIEW_CHECK_DAA") != nullptr;
	int64_t daa_mismatches = 0;
	std::vector<int8_t> seq, frames[6];
	for (;; ++qnum) {
		if (pos + 4 > aln_end) throw std::runtime_error("Truncated DAA file.");
		uint32_t size;
		std::memcpy(&size, file.data() + pos, 4);
		pos += 4;
		if (size == 0) break;
		if (pos + size > aln_end) throw std::runtime_error("Truncated DAA file.");
		const uint8_t* r = file.data() + pos;
		const uint8_t* const rend = r + size;
		pos += size;
		// DAA_query_record::init (daa_record.cpp:30-50): length, name, flags, packed letters
		uint32_t qlen;
		std::memcpy(&qlen, r, 4);
		const char* qname = (const char*)r + 4;
		const size_t name_len = strnlen(qname, (size_t)(rend - r) - 4);
		const uint8_t* q = r + 4 + name_len + 1;
		if (q >= rend) throw std::runtime_error("Malformed DAA query record.");
		const unsigned bits = blastx ? ((*q & 1) ? 3 : 2) : 5;
		++q;
		const size_t packed = ((size_t)qlen * bits + 7) / 8;
		if (q + packed > rend) throw std::runtime_error("Malformed DAA query record.");
		seq.assign(qlen, 0);
		{
			unsigned x = 0, n = 0;
			size_t l = 0;
			for (size_t i = 0; i < packed; ++i) {
				x |= (unsigned)q[i] << n;
				n += 8;
				while (n >= bits && l < qlen) { seq[l++] = (int8_t)(x & ((1u << bits) - 1)); n -= bits; x >>= bits; }
			}
		}
		q += packed;
		int32_t flen[6] = { (int32_t)qlen, 0, 0, 0, 0, 0 };
		if (blastx) {
			int8_t* outp[6];
			for (int k = 0; k < 6; ++k) { frames[k].assign(qlen / 3 + 1, 0); outp[k] = frames[k].data(); }
			// Translator::translate: the plain six-frame translation, no ORF masking (min_orf 1)
			if (dmnd_translate_opts(seq.data(), (int32_t)qlen, o.gencode, 3, 1, outp, flen) != DMND_OK) throw std::runtime_error(dmnd_last_error());
		}
		const std::string qtitle(qname, name_len);
		bool intro = false;
		uint32_t last_subject = UINT32_MAX;
		int32_t hit_num = -1, hsp_num = 0, top_score = 0;
		while (q < rend) {
			uint32_t dict = 0;
			dmnd_match m;
			int64_t tr_off = 0, usedb = 0;
			if (dmnd_daa_match_read(q, (int64_t)(rend - q), blastx ? 1 : 0, (int32_t)qlen, &dict, &m, &tr_off, &usedb) != DMND_OK) throw std::runtime_error(dmnd_last_error());
			const uint8_t* tr = q + tr_off;
			q += usedb;
			if (dict >= used) throw std::runtime_error("Malformed DAA match record.");
			if (dict == last_subject) ++hsp_num; else { hsp_num = 0; ++hit_num; last_subject = dict; }
			if (hit_num == 0 && hsp_num == 0) top_score = m.hsp.score;
			if (m.frame > 2 && o.forwardonly) continue;
			// Config::output_range (basic/config.h:426-432)
			if (o.top >= 0.0 ? !((1.0 - (double)m.hsp.score / top_score) * 100 <= o.top) : !(hit_num < K)) break;
			uint32_t slen;
			std::memcpy(&slen, ref_len + (size_t)dict * 4, 4);
			const int8_t* ctx = blastx ? frames[m.frame].data() : seq.data();
			const int32_t ctx_len = blastx ? flen[m.frame] : (int32_t)qlen;
			m.query = (uint32_t)qnum; m.target = dict;
			m.hsp.transcript_off = 0;
			if (blastx) {
				const int strand = m.frame / 3;
				const int8_t* f3[3] = { frames[strand * 3].data(), frames[strand * 3 + 1].data(), frames[strand * 3 + 2].data() };
				const int32_t l3[3] = { flen[strand * 3], flen[strand * 3 + 1], flen[strand * 3 + 2] };
				if (dmnd_hsp_from_transcript_frames(&p, f3, l3, (int32_t)qlen, flen[0], (int32_t)slen, tr, &m) != DMND_OK) throw std::runtime_error(dmnd_last_error());
			}
			else if (dmnd_hsp_from_transcript(&p, ctx, ctx_len, flen[0], (int32_t)slen, tr, &m) != DMND_OK) throw std::runtime_error(dmnd_last_error());
			dmnd_hsp_view v;
			v.match = &m; v.transcript = tr; v.qtitle = qtitle.c_str(); v.stitle = ref_name[dict]; v.qseq = ctx; v.qlen = ctx_len; v.slen = (int32_t)slen; v.full_sseq = nullptr;
			v.source_seq = blastx ? seq.data() : nullptr; v.source_len = blastx ? (int32_t)qlen : 0; v.qnum = 0; v.snum = (int64_t)dict;      // the reference's view hands 0 to every record as the query's ordinal (daa_record.h:46-52)
			for (int k = 0; k < 3; ++k) v.qframes[k] = blastx ? frames[(m.frame / 3) * 3 + k].data() : nullptr;
			if (check_daa) {
				// DMND_VIEW_CHECK_DAA=1 (tests): the record written again from what was read must be the bytes that were read
				std::vector<char> again((size_t)usedb + 64);
				const int64_t w = dmnd_format_daa_match(&v, dict, again.data(), (int64_t)again.size());
				if (w != usedb || std::memcmp(again.data(), q - usedb, (size_t)usedb) != 0) ++daa_mismatches;
			}
			big.resize((size_t)m.hsp.length * 8 + (size_t)qlen * 3 + qtitle.size() * 6 + std::strlen(v.stitle) * 6 + 4096);
			if (!intro) {
				intro = true;
				if (fmt == V_XML && qnum == 0)
					put(dmnd_format_xml_header(blastx ? "blastx" : "blastp", ("diamond-hip ABI " + std::to_string(dmnd_abi_version())).c_str(), "", qtitle.c_str(), (int32_t)qlen, matrix,
						gap_open, gap_extend, max_evalue, big.data(), (int64_t)big.size()));
				if (fmt == V_PAIRWISE) put(dmnd_format_pairwise_intro(qtitle.c_str(), (int32_t)qlen, 0, big.data(), (int64_t)big.size()));
				if (fmt == V_XML) put(dmnd_format_xml_query_intro(qtitle.c_str(), qnum, (int32_t)qlen, big.data(), (int64_t)big.size()));
			}
			if (fmt == V_FIELDS) put(dmnd_format_fields(&v, field_ids.data(), (int)field_ids.size(), big.data(), (int64_t)big.size()));
			else if (fmt == V_PAIRWISE) put(dmnd_format_pairwise(&v, p.matrix8, big.data(), (int64_t)big.size()));
			else if (fmt == V_XML) put(dmnd_format_xml(&v, hit_num, hsp_num, p.matrix8, big.data(), (int64_t)big.size()));
			else if (fmt == V_SAM) put(dmnd_format_sam(&v, nullptr, big.data(), (int64_t)big.size()));
			else put(dmnd_format_paf(&v, nullptr, big.data(), (int64_t)big.size()));
			++n_hsps;
		}
		if (fmt == V_XML && intro) put(dmnd_format_xml_query_epilog(0, (int64_t)db_seqs, (int64_t)db_letters, p.K, p.lambda, big.data(), (int64_t)big.size()));
	}
	if (fmt == V_XML) out.write("</BlastOutput_iterations>\n</BlastOutput>");
	out.close();
	std::cerr << "Printed " << n_hsps << " HSPs of " << qnum << " queries.\n";
	if (check_daa) { std::cerr << "DAA records written back: " << daa_mismatches << " differ.\n"; if (daa_mismatches) return 1; }
	return 0;
}

}  // namespace

int main(int argc, char** argv)
{
	// Copies through blit kernels, not the SDMA engines (unless the caller chose otherwise): the driver also runs its page-table
	// updates on SDMA, and after the GBs of scratch a masking call maps and unmaps, a half-megabyte copy of seed hits queued behind
	// them for 10-27 ms (round 3 timeline); block uploads run at the same 18 GB/s either way. Must be set before the runtime starts.
	::setenv("HSA_ENABLE_SDMA", "0", 0);
	try {
		const Options o = parse(argc, argv);
		if (o.command == "version") { std::cout << "diamond-hip (MI355X back end of DIAMOND's seed-and-extend path), ABI " << dmnd_abi_version() << "\n"; return 0; }
		if (o.command == "help" || o.command == "--help") {
			std::cout << "Syntax: diamond-hip COMMAND [OPTIONS]\n"
				"  makedb --in FASTA -d DB              build a .dmnd database\n"
				"  blastp -q PROTEINS -d DB -o OUT      protein search;  blastx -q READS -d DB -o OUT   translated search\n"
				"  view -a ARCHIVE.daa -o OUT [-f ...]    print a DAA archive in another format\n  dbinfo -d DB                         sequences, letters and format of a .dmnd file\n  version\n\n"
				"input        -q / -d: FASTA or FASTQ, gzip-compressed or not; -d also a .dmnd file\n"
				"sensitivity  --fast | (none: default mode) | --mid-sensitive | --sensitive | --more-sensitive | --very-sensitive | --ultra-sensitive; --shapes N\n"
				"scoring      --matrix BLOSUM45|50|62|80|90|PAM30|70|250  --gapopen N  --gapextend N  --comp-based-stats 0|1\n"
				"masking      --masking tantan|seg|none  --motif-masking 0|1\n"
				"extension    --ext banded-fast|banded-slow|full\n"
				"reporting    -k N  --max-hsps N  --global-ranking N  --top PCT  -e EVALUE  --min-score BITS  --id PCT  --query-cover PCT  --subject-cover PCT  --no-self-hits\n"
				"             --unal 0|1  --un FILE  --al FILE  --header [simple|verbose]  --compress 1  --salltitles  --sallseqid\n"
				"formats      -f 6 [FIELD...] | 0 (pairwise) | 5 (XML) | 100 (DAA) | 101 (SAM) | 103 (PAF)\n"
				"translated   --strand both|plus|minus  --query-gencode N  --min-orf N\n"
				"expert       --dbsize LETTERS  --id2 N  --seed-cut X  --gapped-filter-evalue E  --stop-match-score N\n"
				"resources    -p THREADS  --gpus N  -b BLOCK_SIZE  -c INDEX_CHUNKS  --algo 0|1|auto\n";
			return 0;
		}
		if (o.command == "makedb") {
			if (o.db.empty()) throw std::runtime_error("Missing parameter: database file (--db/-d)");        // no --in: the sequences come from standard input
			SeqBlock b;
			read_fasta(o.in, b);
			write_dmnd(o.db, b);
			return 0;
		}
		if (dmnd_set_format_flags(o.format_flags) != DMND_OK) throw std::runtime_error(dmnd_last_error());
		if (o.command == "dbinfo") {                                                   // db_info, data/sequence_file.cpp:876-890
			if (o.db.empty()) throw std::runtime_error("Missing parameter: database file (--db/-d)");
			std::string path = o.db;
			if (!std::ifstream(path).good() && std::ifstream(path + ".dmnd").good()) path += ".dmnd";
			std::ifstream f(path, std::ios::binary);
			if (!f) throw std::runtime_error("Error opening file " + path);
			uint64_t magic = 0, sequences = 0, letters = 0; uint32_t build = 0, version = 0;
			f.read((char*)&magic, 8); f.read((char*)&build, 4); f.read((char*)&version, 4); f.read((char*)&sequences, 8); f.read((char*)&letters, 8);
			if (!f || magic != DMND_MAGIC) throw std::runtime_error("Database file is not a DIAMOND database.");
			std::printf("%23s  %s\n%23s  %u\n%23s  %u\n%23s  %llu\n%23s  %llu\n", "Database type", "Diamond database", "Database format version", version, "Diamond build", build,
				"Sequences", (unsigned long long)sequences, "Letters", (unsigned long long)letters);
			return 0;
		}
		if (o.command == "view") return run_view(o);
		if ((o.command == "blastp" || o.command == "blastx") && !o.daa.empty()) {       // -a FILE: the legacy spelling of -f 100 -o FILE (basic/config.cpp:716-724)
			if (!o.out.empty()) throw std::runtime_error("Options --daa and --out cannot be used together.");
			if (!o.outfmt.empty() && o.outfmt[0] != "daa" && o.outfmt[0] != "100") throw std::runtime_error("Invalid parameter: --daa/-a. Output file is specified with the --out/-o parameter.");
			Options legacy = o;
			legacy.outfmt.assign(1, "100");
			legacy.out = o.daa;
			return run_blastp(legacy);
		}
		if (o.command == "blastp" || o.command == "blastx") return run_blastp(o);
		throw std::runtime_error("Invalid command: " + o.command + " (makedb, blastp, blastx, view and dbinfo are part of this build)");
	}
	catch (const std::exception& e) {
		std::cerr << "Error: " << e.what() << std::endl;          // main.cpp:211-232
		return 1;
	}
}
