// blow5_reader.cpp - host-side BLOW5 reader that hands out the COMPRESSED signal fields.
//
// The reference's loader (slow5lib: slow5_open slow5.h:345, slow5_get_next :446) inflates a
// record and decodes its signal on the CPU; the benchmark then compresses the int16 samples
// again.  This reader stops one step earlier: it parses the file and record framing
// (slow5.c:787-870 header, :3903-3965 record layout), inflates the record if the file uses
// record compression, and returns the signal field as it is stored - for signal method
// "svb-zd" that is exactly the stream PRESS_HIP_SLOW5_SVB_ZD decodes on the device, so a read
// crosses PCIe at ~1.25 bytes per sample instead of 2.
//
// File:   "BLOW5\1" | major minor patch (u8 each) | record method u8 | num read groups u32 |
//         signal method u8 (version >= 0.2.0) | ... | at byte 64: u32 header size | header text |
//         records | "5WOLB"
// Record: u64 size | bytes (zlib / zstd stream of, or plainly:) u16 read_id_len | read_id |
//         u32 read_group | f64 digitisation, offset, range, sampling_rate | u64 len_raw_signal |
//         signal | auxiliary fields.   len_raw_signal counts BYTES when the signal is
//         compressed (slow5.c:3960) and samples when it is not.
// The writer is the other direction (slow5_write / slow5_rec_to_mem, slow5.c:3903-4010): it copies
// the source file's header, takes records whose signal field the caller has replaced (e.g. by the
// device's svb-zd encoding) and frames them - so a compressor can emit BLOW5 directly.
// Host code only: no HIP call in this file.

#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <deque>
#include <string>
#include <thread>
#include <vector>

#include "../../include/press_hip.h"

namespace {

thread_local char b5_err[256];

int b5_fail(int code, const char *msg)
{
	snprintf(b5_err, sizeof b5_err, "%s", msg);
	return code;
}

// zlib / zstd are looked up at run time: the library must load on a box without them
struct Inflaters {
	void *hz = nullptr, *hs = nullptr;
	int (*z_uncompress)(unsigned char *, unsigned long *, const unsigned char *, unsigned long) = nullptr;
	int (*z_compress2)(unsigned char *, unsigned long *, const unsigned char *, unsigned long, int) = nullptr;
	unsigned long (*z_bound)(unsigned long) = nullptr;
	size_t (*zs_decompress)(void *, size_t, const void *, size_t) = nullptr;
	unsigned long long (*zs_content_size)(const void *, size_t) = nullptr;
	unsigned (*zs_is_error)(size_t) = nullptr;
	bool tried = false;
	void open()
	{
		if (tried)
			return;
		tried = true;
		for (const char *n : { "libz.so.1", "libz.so" }) {
			if ((hz = dlopen(n, RTLD_NOW | RTLD_LOCAL)))
				break;
		}
		if (hz) {
			z_uncompress = (decltype(z_uncompress)) dlsym(hz, "uncompress");
			z_compress2 = (decltype(z_compress2)) dlsym(hz, "compress2");
			z_bound = (decltype(z_bound)) dlsym(hz, "compressBound");
		}
		// (what the process already holds first: two copies of libzstd in one process free each other's memory)
		if (!(hs = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD))) {
			for (const char *n : { "libzstd.so.1", "libzstd.so" }) {
				if ((hs = dlopen(n, RTLD_NOW | RTLD_LOCAL)))
					break;
			}
		}
		if (!hs)
			hs = dlopen("/opt/conda/lib/libzstd.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
		if (hs) {
			zs_decompress = (decltype(zs_decompress)) dlsym(hs, "ZSTD_decompress");
			zs_content_size = (decltype(zs_content_size)) dlsym(hs, "ZSTD_getFrameContentSize");
			zs_is_error = (decltype(zs_is_error)) dlsym(hs, "ZSTD_isError");
		}
	}
} inf;

} // namespace

struct press_hip_blow5 {
	FILE *fp = nullptr;
	int record_method = 0, signal_method = 0;
	uint8_t version[3] = { 0, 0, 0 };
	uint32_t num_read_groups = 0;
	std::string header;
	uint8_t fixed[68];              // the 64-byte binary header + the text header's size
	std::vector<uint8_t> rec;       // the record at the head of the queue (inflated)
	bool have_pending = false;      // `rec` holds a record that has not been handed out yet
	bool at_eof = false;
	// records are read from the file in blocks and inflated by a pool of host threads (zlib's inflate of one record
	// after the other on one thread fed the device at ~0.3 GB/s): what is inflated waits here in file order
	std::deque<std::vector<uint8_t>> ready;
	int threads = 0;                // 0: as many as the host offers, at most 32
};

namespace {

// inflate one record as stored (`comp`) into `rec`; 0 ok, < 0 error (message in *err)
int inflate_record(int record_method, std::vector<uint8_t> &comp, std::vector<uint8_t> &rec, const char **err)
{
	const size_t n = comp.size();
	if (record_method == 0) {
		rec.swap(comp);
		return 0;
	}
	if (record_method == 1) { // zlib (slow5_press.c ptr_depress_zlib): the inflated size is not stored
		size_t cap = n * 4 + 4096;
		for (;;) {
			rec.resize(cap);
			unsigned long out = (unsigned long) cap;
			const int rc = inf.z_uncompress(rec.data(), &out, comp.data(), (unsigned long) n);
			if (rc == 0) {
				rec.resize(out);
				return 0;
			}
			if (rc != -5 /* Z_BUF_ERROR */ || cap > (1ull << 34)) {
				*err = "BLOW5: zlib could not inflate a record";
				return PRESS_HIP_EARG;
			}
			cap *= 2;
		}
	}
	// zstd
	const unsigned long long want = inf.zs_content_size(comp.data(), n);
	if (want == 0ULL - 1 || want == 0ULL - 2 || want > (1ull << 34)) {
		*err = "BLOW5: zstd frame without a usable content size";
		return PRESS_HIP_EARG;
	}
	rec.resize((size_t) want);
	const size_t out = inf.zs_decompress(rec.data(), (size_t) want, comp.data(), n);
	if (inf.zs_is_error(out) || out != want) {
		*err = "BLOW5: zstd could not inflate a record";
		return PRESS_HIP_EARG;
	}
	return 0;
}

unsigned pool_size(int asked, size_t jobs)
{
	unsigned t = asked > 0 ? (unsigned) asked : std::thread::hardware_concurrency();
	if (t == 0)
		t = 1;
	if (t > 32)
		t = 32;
	if (t > jobs)
		t = (unsigned) jobs;
	return t ? t : 1;
}

// Reads the next block of records from the file (up to `want` records or ~64 MiB as stored) and inflates them on the
// pool, into f->ready in file order.  0 ok (f->ready may stay empty at the end of the file), < 0 error.
int fill_ready(press_hip_blow5 *f, uint32_t want)
{
	std::vector<std::vector<uint8_t>> comp;
	uint64_t bytes = 0;
	while (!f->at_eof && comp.size() < want && bytes < (64ull << 20)) {
		uint8_t sz[8];
		const size_t got = fread(sz, 1, 8, f->fp);
		if (got >= 5 && !memcmp(sz, "5WOLB", 5)) {
			f->at_eof = true;
			break;
		}
		if (got == 0 && feof(f->fp)) { // a file without the end marker: tolerated like a truncated tail is not
			f->at_eof = true;
			break;
		}
		if (got != 8)
			return b5_fail(PRESS_HIP_EARG, "BLOW5: truncated record size");
		uint64_t n;
		memcpy(&n, sz, 8);
		if (n == 0 || n > (1ull << 34))
			return b5_fail(PRESS_HIP_EARG, "BLOW5: implausible record size");
		comp.emplace_back(n);
		if (fread(comp.back().data(), 1, n, f->fp) != n)
			return b5_fail(PRESS_HIP_EARG, "BLOW5: truncated record");
		bytes += n;
	}
	if (comp.empty())
		return 0;
	if (f->record_method) {
		inf.open();
		if (f->record_method == 1 && !inf.z_uncompress)
			return b5_fail(PRESS_HIP_EARG, "BLOW5: zlib record compression but libz is not available");
		if (f->record_method == 2 && (!inf.zs_decompress || !inf.zs_content_size || !inf.zs_is_error))
			return b5_fail(PRESS_HIP_EARG, "BLOW5: zstd record compression but libzstd is not available");
		if (f->record_method > 2)
			return b5_fail(PRESS_HIP_EARG, "BLOW5: unknown record compression method");
	}
	std::vector<std::vector<uint8_t>> out(comp.size());
	std::atomic<size_t> next(0);
	std::atomic<int> bad(0);
	const char *first_err = nullptr;
	auto work = [&]() {
		for (;;) {
			const size_t i = next.fetch_add(1);
			if (i >= comp.size())
				break;
			const char *e = nullptr;
			if (inflate_record(f->record_method, comp[i], out[i], &e) && !bad.exchange(1))
				first_err = e;
		}
	};
	const unsigned nt = f->record_method ? pool_size(f->threads, comp.size()) : 1;
	std::vector<std::thread> pool;
	for (unsigned t = 1; t < nt; t++)
		pool.emplace_back(work);
	work();
	for (auto &t : pool)
		t.join();
	if (bad.load())
		return b5_fail(PRESS_HIP_EARG, first_err ? first_err : "BLOW5: a record does not inflate");
	for (auto &r : out)
		f->ready.emplace_back(std::move(r));
	return 0;
}

// the next record into f->rec (inflated); 0 ok, 1 end of file, < 0 error.  `ahead`: records the caller may still take
int next_record(press_hip_blow5 *f, uint32_t ahead = 1)
{
	if (f->ready.empty()) {
		const int rc = fill_ready(f, ahead < 16 ? 16 : ahead);
		if (rc)
			return rc;
		if (f->ready.empty())
			return 1;
	}
	f->rec.swap(f->ready.front());
	f->ready.pop_front();
	return 0;
}

struct RecView {
	const uint8_t *id;
	uint32_t id_len;
	const uint8_t *sig;
	uint64_t sig_bytes;
	uint64_t sig_pos; // offset of the signal field in the inflated record (its u64 length sits 8 bytes before)
	uint32_t nsamples;
};

int parse_record(const press_hip_blow5 *f, RecView *v)
{
	const std::vector<uint8_t> &r = f->rec;
	size_t p = 0;
	if (r.size() < 2)
		return b5_fail(PRESS_HIP_EARG, "BLOW5: record too short");
	uint16_t idl;
	memcpy(&idl, r.data(), 2);
	p = 2;
	if (r.size() < p + idl + 4 + 32 + 8)
		return b5_fail(PRESS_HIP_EARG, "BLOW5: record too short");
	v->id = r.data() + p;
	v->id_len = idl;
	p += (size_t) idl + 4 + 32;
	uint64_t len;
	memcpy(&len, r.data() + p, 8);
	p += 8;
	const uint64_t bytes = f->signal_method ? len : len * 2;
	if (bytes > r.size() - p)
		return b5_fail(PRESS_HIP_EARG, "BLOW5: signal field runs past the record");
	v->sig = r.data() + p;
	v->sig_pos = p;
	v->sig_bytes = bytes;
	if (f->signal_method) {
		if (bytes < 4)
			return b5_fail(PRESS_HIP_EARG, "BLOW5: svb-zd signal without its count");
		memcpy(&v->nsamples, v->sig, 4);
	} else {
		if (len > 0xFFFFFFFFull)
			return b5_fail(PRESS_HIP_EARG, "BLOW5: signal too long");
		v->nsamples = (uint32_t) len;
	}
	return 0;
}

} // namespace

extern "C" {

const char *press_hip_blow5_last_error(void) { return b5_err; }

int press_hip_blow5_open(const char *path, press_hip_blow5 **out)
{
	if (!path || !out)
		return b5_fail(PRESS_HIP_EARG, "BLOW5: NULL argument");
	FILE *fp = fopen(path, "rb");
	if (!fp)
		return b5_fail(PRESS_HIP_EARG, "BLOW5: cannot open the file");
	uint8_t h[68];
	if (fread(h, 1, sizeof h, fp) != sizeof h || memcmp(h, "BLOW5\1", 6)) {
		fclose(fp);
		return b5_fail(PRESS_HIP_EARG, "BLOW5: bad magic number");
	}
	press_hip_blow5 *f = new press_hip_blow5;
	f->fp = fp;
	memcpy(f->fixed, h, sizeof h);
	memcpy(f->version, h + 6, 3);
	f->record_method = h[9];
	memcpy(&f->num_read_groups, h + 10, 4);
	// slow5.c:820: the signal method byte exists from version 0.2.0 on
	const bool has_sig = f->version[0] > 0 || f->version[1] >= 2;
	f->signal_method = has_sig ? h[14] : 0;
	uint32_t hs;
	memcpy(&hs, h + 64, 4);
	if (f->record_method > 2 || f->signal_method > 1 || hs > (1u << 30)) {
		press_hip_blow5_close(f);
		return b5_fail(PRESS_HIP_EARG, "BLOW5: unsupported compression method or header size");
	}
	f->header.resize(hs);
	if (hs && fread(&f->header[0], 1, hs, fp) != hs) {
		press_hip_blow5_close(f);
		return b5_fail(PRESS_HIP_EARG, "BLOW5: truncated header");
	}
	*out = f;
	return 0;
}

void press_hip_blow5_close(press_hip_blow5 *f)
{
	if (!f)
		return;
	if (f->fp)
		fclose(f->fp);
	delete f;
}

int press_hip_blow5_threads(press_hip_blow5 *f, int threads)
{
	if (!f || threads < 0)
		return b5_fail(PRESS_HIP_EARG, "BLOW5: bad argument");
	f->threads = threads;
	return 0;
}

int press_hip_blow5_methods(const press_hip_blow5 *f, int *record_method, int *signal_method)
{
	if (!f)
		return b5_fail(PRESS_HIP_EARG, "BLOW5: NULL handle");
	if (record_method)
		*record_method = f->record_method;
	if (signal_method)
		*signal_method = f->signal_method;
	return 0;
}

int press_hip_blow5_next(press_hip_blow5 *f, uint32_t max_reads, uint8_t *arena, uint64_t arena_cap,
			 uint64_t *sig_off, uint64_t *sig_len, uint32_t *n_samples, char *read_ids, uint32_t *got)
{
	if (!f || !arena || !sig_off || !sig_len || !n_samples || !got)
		return b5_fail(PRESS_HIP_EARG, "BLOW5: NULL argument");
	uint32_t k = 0;
	uint64_t used = 0;
	while (k < max_reads) {
		if (!f->have_pending) {
			if (f->at_eof && f->ready.empty())
				break;
			const int rc = next_record(f, max_reads - k);
			if (rc == 1)
				break;
			if (rc)
				return rc;
		}
		f->have_pending = true;
		RecView v;
		int rc = parse_record(f, &v);
		if (rc)
			return rc;
		const uint64_t at = (used + 15) & ~15ull; // every field starts on a 16-byte boundary
		if (at + v.sig_bytes > arena_cap) {
			if (k == 0)
				return b5_fail(PRESS_HIP_EARG, "BLOW5: the arena cannot hold even one signal");
			break; // this record opens the next batch
		}
		memcpy(arena + at, v.sig, v.sig_bytes);
		sig_off[k] = at;
		sig_len[k] = v.sig_bytes;
		n_samples[k] = v.nsamples;
		if (read_ids) {
			char *dst = read_ids + (size_t) k * PRESS_HIP_BLOW5_ID_LEN;
			const uint32_t c = v.id_len < PRESS_HIP_BLOW5_ID_LEN - 1 ? v.id_len : PRESS_HIP_BLOW5_ID_LEN - 1;
			memcpy(dst, v.id, c);
			memset(dst + c, 0, PRESS_HIP_BLOW5_ID_LEN - c);
		}
		used = at + v.sig_bytes;
		f->have_pending = false;
		k++;
	}
	*got = k;
	return 0;
}


// Whole (inflated) records for a transcoder: as press_hip_blow5_next, but the arena receives the
// records themselves; sig_pos / sig_len locate the signal field inside record k (the u64 in front
// of it is its length field: bytes when the file compresses signals, samples when it does not).
int press_hip_blow5_next_records(press_hip_blow5 *f, uint32_t max_reads, uint8_t *arena, uint64_t arena_cap,
				 uint64_t *rec_off, uint64_t *rec_len, uint64_t *sig_pos, uint64_t *sig_len,
				 uint32_t *n_samples, uint32_t *got)
{
	if (!f || !arena || !rec_off || !rec_len || !sig_pos || !sig_len || !n_samples || !got)
		return b5_fail(PRESS_HIP_EARG, "BLOW5: NULL argument");
	uint32_t k = 0;
	uint64_t used = 0;
	while (k < max_reads) {
		if (!f->have_pending) {
			if (f->at_eof && f->ready.empty())
				break;
			const int rc = next_record(f, max_reads - k);
			if (rc == 1)
				break;
			if (rc)
				return rc;
		}
		f->have_pending = true;
		RecView v;
		int rc = parse_record(f, &v);
		if (rc)
			return rc;
		const uint64_t at = (used + 15) & ~15ull;
		if (at + f->rec.size() > arena_cap) {
			if (k == 0)
				return b5_fail(PRESS_HIP_EARG, "BLOW5: the arena cannot hold even one record");
			break;
		}
		memcpy(arena + at, f->rec.data(), f->rec.size());
		rec_off[k] = at;
		rec_len[k] = f->rec.size();
		sig_pos[k] = v.sig_pos;
		sig_len[k] = v.sig_bytes;
		n_samples[k] = v.nsamples;
		used = at + f->rec.size();
		f->have_pending = false;
		k++;
	}
	*got = k;
	return 0;
}

struct press_hip_blow5_writer {
	FILE *fp = nullptr;
	int record_method = 0, signal_method = 0;
	std::vector<uint8_t> rec, comp;
	// the index slow5lib keeps beside a BLOW5 file (slow5_idx.c:269 slow5_idx_write): read id -> offset and size of
	// its record (the size field included); written by press_hip_blow5_finish when asked for
	bool want_index = false;
	std::string path;
	uint8_t version[3] = { 0, 2, 0 };
	uint64_t at = 0; // file offset of the next record
	struct Entry {
		std::string id;
		uint64_t offset, size;
	};
	std::vector<Entry> index;
	int threads = 0;
};

int press_hip_blow5_create(const char *path, const press_hip_blow5 *like, int record_method, int signal_method,
			   press_hip_blow5_writer **out)
{
	if (!path || !like || !out)
		return b5_fail(PRESS_HIP_EARG, "BLOW5: NULL argument");
	if (record_method < 0 || record_method > 1 || signal_method < 0 || signal_method > 1)
		return b5_fail(PRESS_HIP_EARG, "BLOW5 writer: record method none / zlib, signal method none / svb-zd");
	if (record_method == 1) {
		inf.open();
		if (!inf.z_compress2 || !inf.z_bound)
			return b5_fail(PRESS_HIP_EARG, "BLOW5 writer: zlib record compression but libz is not available");
	}
	FILE *fp = fopen(path, "wb");
	if (!fp)
		return b5_fail(PRESS_HIP_EARG, "BLOW5 writer: cannot create the file");
	uint8_t h[68];
	memcpy(h, like->fixed, sizeof h);
	// the signal method byte exists from 0.2.0 on (slow5.c:4640): lift older headers
	if (h[6] == 0 && h[7] < 2) {
		h[7] = 2;
		h[8] = 0;
	}
	h[9] = (uint8_t) record_method;
	h[14] = (uint8_t) signal_method;
	if (fwrite(h, 1, sizeof h, fp) != sizeof h ||
	    (like->header.size() && fwrite(like->header.data(), 1, like->header.size(), fp) != like->header.size())) {
		fclose(fp);
		return b5_fail(PRESS_HIP_EARG, "BLOW5 writer: write failed");
	}
	press_hip_blow5_writer *w = new press_hip_blow5_writer;
	w->fp = fp;
	w->record_method = record_method;
	w->signal_method = signal_method;
	w->path = path;
	memcpy(w->version, h + 6, 3);
	w->at = sizeof h + like->header.size();
	*out = w;
	return 0;
}

// Keep an index while writing and leave it as <path>.idx when the file is finished: slow5lib's slow5_idx_load /
// slow5_get then find a read of the transcoded file by its id without scanning it (slow5_idx.c).
int press_hip_blow5_index(press_hip_blow5_writer *w, int enable)
{
	if (!w)
		return b5_fail(PRESS_HIP_EARG, "BLOW5 writer: NULL handle");
	w->want_index = enable != 0;
	return 0;
}

namespace {

// frame one record (pre | u64 length | sig | post) and, for zlib files, deflate it: -> body
int frame_record(int record_method, int signal_method, const uint8_t *pre, uint64_t pre_len, const uint8_t *sig, uint64_t sig_len,
		 const uint8_t *post, uint64_t post_len, std::vector<uint8_t> &rec, std::vector<uint8_t> &body)
{
	const uint64_t lenfield = signal_method ? sig_len : sig_len / 2; // slow5.c:3960
	rec.resize(pre_len + 8 + sig_len + post_len);
	memcpy(rec.data(), pre, pre_len);
	memcpy(rec.data() + pre_len, &lenfield, 8);
	if (sig_len)
		memcpy(rec.data() + pre_len + 8, sig, sig_len);
	if (post_len)
		memcpy(rec.data() + pre_len + 8 + sig_len, post, post_len);
	if (record_method == 1) {
		unsigned long cap = inf.z_bound((unsigned long) rec.size());
		body.resize(cap);
		if (inf.z_compress2(body.data(), &cap, rec.data(), (unsigned long) rec.size(), -1 /* Z_DEFAULT_COMPRESSION */))
			return -1;
		body.resize(cap);
	} else {
		body.swap(rec);
	}
	return 0;
}

int put_record(press_hip_blow5_writer *w, const uint8_t *pre, uint64_t pre_len, const std::vector<uint8_t> &body)
{
	const uint64_t n = body.size();
	if (fwrite(&n, 8, 1, w->fp) != 1 || fwrite(body.data(), 1, n, w->fp) != n)
		return b5_fail(PRESS_HIP_EARG, "BLOW5 writer: write failed");
	if (w->want_index) {
		uint16_t idl = 0;
		if (pre_len >= 2)
			memcpy(&idl, pre, 2);
		if ((uint64_t) idl + 2 > pre_len)
			return b5_fail(PRESS_HIP_EARG, "BLOW5 writer: the record's read id runs past its front part");
		w->index.push_back({ std::string((const char *) pre + 2, idl), w->at, 8 + n });
	}
	w->at += 8 + n;
	return 0;
}

} // namespace


// One record: `pre` = everything in front of the signal's length field (read id ... sampling
// rate), `sig` = the signal field in the writer's signal method, `post` = the auxiliary fields.
int press_hip_blow5_write(press_hip_blow5_writer *w, const uint8_t *pre, uint64_t pre_len, const uint8_t *sig,
			  uint64_t sig_len, const uint8_t *post, uint64_t post_len)
{
	if (!w || !pre || (!sig && sig_len) || (!post && post_len))
		return b5_fail(PRESS_HIP_EARG, "BLOW5 writer: NULL argument");
	if (frame_record(w->record_method, w->signal_method, pre, pre_len, sig, sig_len, post, post_len, w->rec, w->comp))
		return b5_fail(PRESS_HIP_EARG, "BLOW5 writer: zlib failed");
	return put_record(w, pre, pre_len, w->comp);
}

// n records at once: framed and deflated by a pool of host threads (zlib's deflate is the slow part of writing a
// BLOW5: ~30 MB/s on one thread), written in the order given.
int press_hip_blow5_write_batch(press_hip_blow5_writer *w, uint32_t n, const uint8_t *const *pre, const uint64_t *pre_len,
				const uint8_t *const *sig, const uint64_t *sig_len, const uint8_t *const *post,
				const uint64_t *post_len)
{
	if (!w || (n && (!pre || !pre_len || !sig || !sig_len || !post_len)))
		return b5_fail(PRESS_HIP_EARG, "BLOW5 writer: NULL argument");
	std::vector<std::vector<uint8_t>> body(n);
	std::atomic<uint32_t> next(0);
	std::atomic<int> bad(0);
	auto work = [&]() {
		std::vector<uint8_t> rec;
		for (;;) {
			const uint32_t i = next.fetch_add(1);
			if (i >= n)
				break;
			if (frame_record(w->record_method, w->signal_method, pre[i], pre_len[i], sig[i], sig_len[i],
					 post ? post[i] : nullptr, post_len[i], rec, body[i]))
				bad.store(1);
		}
	};
	const unsigned nt = w->record_method ? pool_size(w->threads, n) : 1;
	std::vector<std::thread> pool;
	for (unsigned t = 1; t < nt; t++)
		pool.emplace_back(work);
	work();
	for (auto &t : pool)
		t.join();
	if (bad.load())
		return b5_fail(PRESS_HIP_EARG, "BLOW5 writer: zlib failed");
	for (uint32_t i = 0; i < n; i++) {
		const int rc = put_record(w, pre[i], pre_len[i], body[i]);
		if (rc)
			return rc;
	}
	return 0;
}

int press_hip_blow5_finish(press_hip_blow5_writer *w)
{
	if (!w)
		return 0;
	int rc = 0;
	if (w->fp) {
		if (fwrite("5WOLB", 1, 5, w->fp) != 5)
			rc = b5_fail(PRESS_HIP_EARG, "BLOW5 writer: write failed");
		if (fclose(w->fp))
			rc = b5_fail(PRESS_HIP_EARG, "BLOW5 writer: close failed");
	}
	if (!rc && w->want_index) {
		// slow5_idx.c:269: "SLOW5IDX\1" | the file's version | zeros up to byte 64 | per read: u16 id length, id,
		// u64 offset, u64 size | "XDI5WOLS"
		FILE *ip = fopen((w->path + ".idx").c_str(), "wb");
		uint8_t head[64] = { 'S', 'L', 'O', 'W', '5', 'I', 'D', 'X', 1, w->version[0], w->version[1], w->version[2] };
		bool ok = ip && fwrite(head, 1, sizeof head, ip) == sizeof head;
		for (size_t i = 0; ok && i < w->index.size(); i++) {
			const press_hip_blow5_writer::Entry &e = w->index[i];
			const uint16_t idl = (uint16_t) e.id.size();
			ok = fwrite(&idl, 2, 1, ip) == 1 && (idl == 0 || fwrite(e.id.data(), 1, idl, ip) == idl) &&
			     fwrite(&e.offset, 8, 1, ip) == 1 && fwrite(&e.size, 8, 1, ip) == 1;
		}
		ok = ok && fwrite("XDI5WOLS", 1, 8, ip) == 8;
		if (ip && fclose(ip))
			ok = false;
		if (!ok)
			rc = b5_fail(PRESS_HIP_EARG, "BLOW5 writer: could not write the index");
	}
	delete w;
	return rc;
}

} // extern "C"
