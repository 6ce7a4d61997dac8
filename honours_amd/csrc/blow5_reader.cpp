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

#include <string>
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
	std::vector<uint8_t> comp, rec; // one record: as stored, inflated
	bool have_pending = false;      // `rec` holds a record that did not fit the last batch
	bool at_eof = false;
};

namespace {

// reads the next record into f->rec (inflated); 0 ok, 1 end of file, < 0 error
int next_record(press_hip_blow5 *f)
{
	uint8_t sz[8];
	const size_t got = fread(sz, 1, 8, f->fp);
	if (got >= 5 && !memcmp(sz, "5WOLB", 5)) {
		f->at_eof = true;
		return 1;
	}
	if (got == 0 && feof(f->fp)) { // a file without the end marker: tolerated like a truncated tail is not
		f->at_eof = true;
		return 1;
	}
	if (got != 8)
		return b5_fail(PRESS_HIP_EARG, "BLOW5: truncated record size");
	uint64_t n;
	memcpy(&n, sz, 8);
	if (n == 0 || n > (1ull << 34))
		return b5_fail(PRESS_HIP_EARG, "BLOW5: implausible record size");
	f->comp.resize(n);
	if (fread(f->comp.data(), 1, n, f->fp) != n)
		return b5_fail(PRESS_HIP_EARG, "BLOW5: truncated record");
	if (f->record_method == 0) {
		f->rec.swap(f->comp);
		return 0;
	}
	inf.open();
	if (f->record_method == 1) { // zlib (slow5_press.c ptr_depress_zlib): the inflated size is not stored
		if (!inf.z_uncompress)
			return b5_fail(PRESS_HIP_EARG, "BLOW5: zlib record compression but libz is not available");
		size_t cap = n * 4 + 4096;
		for (;;) {
			f->rec.resize(cap);
			unsigned long out = (unsigned long) cap;
			const int rc = inf.z_uncompress(f->rec.data(), &out, f->comp.data(), (unsigned long) n);
			if (rc == 0) {
				f->rec.resize(out);
				return 0;
			}
			if (rc != -5 /* Z_BUF_ERROR */ || cap > (1ull << 34))
				return b5_fail(PRESS_HIP_EARG, "BLOW5: zlib could not inflate a record");
			cap *= 2;
		}
	}
	if (f->record_method == 2) { // zstd
		if (!inf.zs_decompress || !inf.zs_content_size || !inf.zs_is_error)
			return b5_fail(PRESS_HIP_EARG, "BLOW5: zstd record compression but libzstd is not available");
		const unsigned long long want = inf.zs_content_size(f->comp.data(), n);
		if (want == 0ULL - 1 || want == 0ULL - 2 || want > (1ull << 34))
			return b5_fail(PRESS_HIP_EARG, "BLOW5: zstd frame without a usable content size");
		f->rec.resize((size_t) want);
		const size_t out = inf.zs_decompress(f->rec.data(), (size_t) want, f->comp.data(), n);
		if (inf.zs_is_error(out) || out != want)
			return b5_fail(PRESS_HIP_EARG, "BLOW5: zstd could not inflate a record");
		return 0;
	}
	return b5_fail(PRESS_HIP_EARG, "BLOW5: unknown record compression method");
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
			if (f->at_eof)
				break;
			const int rc = next_record(f);
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
			if (f->at_eof)
				break;
			const int rc = next_record(f);
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
	*out = w;
	return 0;
}

// One record: `pre` = everything in front of the signal's length field (read id ... sampling
// rate), `sig` = the signal field in the writer's signal method, `post` = the auxiliary fields.
int press_hip_blow5_write(press_hip_blow5_writer *w, const uint8_t *pre, uint64_t pre_len, const uint8_t *sig,
			  uint64_t sig_len, const uint8_t *post, uint64_t post_len)
{
	if (!w || !pre || (!sig && sig_len) || (!post && post_len))
		return b5_fail(PRESS_HIP_EARG, "BLOW5 writer: NULL argument");
	const uint64_t lenfield = w->signal_method ? sig_len : sig_len / 2; // slow5.c:3960
	w->rec.resize(pre_len + 8 + sig_len + post_len);
	memcpy(w->rec.data(), pre, pre_len);
	memcpy(w->rec.data() + pre_len, &lenfield, 8);
	if (sig_len)
		memcpy(w->rec.data() + pre_len + 8, sig, sig_len);
	if (post_len)
		memcpy(w->rec.data() + pre_len + 8 + sig_len, post, post_len);
	const uint8_t *body = w->rec.data();
	uint64_t n = w->rec.size();
	if (w->record_method == 1) {
		unsigned long cap = inf.z_bound((unsigned long) n);
		w->comp.resize(cap);
		if (inf.z_compress2(w->comp.data(), &cap, body, (unsigned long) n, -1 /* Z_DEFAULT_COMPRESSION */))
			return b5_fail(PRESS_HIP_EARG, "BLOW5 writer: zlib failed");
		body = w->comp.data();
		n = cap;
	}
	if (fwrite(&n, 8, 1, w->fp) != 1 || fwrite(body, 1, n, w->fp) != n)
		return b5_fail(PRESS_HIP_EARG, "BLOW5 writer: write failed");
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
	delete w;
	return rc;
}

} // extern "C"
