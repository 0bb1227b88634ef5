// Per-kernel HIP-event timing on the context's stream (used by bench.py for the live roofline numbers).
#include "common.h"
#include <map>
#include <algorithm>

static hipEvent_t takeEvent(bhip_ctx* ctx) {
	if (!ctx->eventPool.empty()) { hipEvent_t e = ctx->eventPool.back(); ctx->eventPool.pop_back(); return e; }
	hipEvent_t e = nullptr;
	(void)hipEventCreate(&e);
	return e;
}

// records kept between two bhip_profile_reset calls (a caller that never resets stops collecting instead of growing without bound)
#define BHIP_PROF_MAX_RECORDS (1 << 18)

ProfScope::ProfScope(bhip_ctx* c, const char* tag, double algBytes, double algFlops) : ctx(c) {
	if (!ctx || !ctx->profiling || ctx->profRecords.size() >= BHIP_PROF_MAX_RECORDS) return;
	ProfRecord r;
	r.tag = tag; r.algBytes = algBytes; r.algFlops = algFlops;
	r.start = takeEvent(ctx);
	r.stop = takeEvent(ctx);
	(void)hipEventRecord(r.start, ctx->stream);
	idx = (int)ctx->profRecords.size();
	ctx->profRecords.push_back(r);
}
ProfScope::~ProfScope() {
	if (idx >= 0) (void)hipEventRecord(ctx->profRecords[idx].stop, ctx->stream);
}

// ctx teardown: every event ever created for this ctx sits in profRecords or in the pool
void bhip_profile_release(bhip_ctx* ctx) {
	for (auto& r : ctx->profRecords) { if (r.start) (void)hipEventDestroy(r.start); if (r.stop) (void)hipEventDestroy(r.stop); }
	ctx->profRecords.clear();
	for (hipEvent_t e : ctx->eventPool) if (e) (void)hipEventDestroy(e);
	ctx->eventPool.clear();
}

struct ProfAgg { double ms = 0, bytes = 0, flops = 0; long long launches = 0; };

extern "C" {

int bhip_profile_enable(bhip_ctx* ctx, int on) {
	if (!ctx) return BHIP_ERR_INVALID;
	ctx->profiling = on != 0;
	return BHIP_OK;
}

int bhip_profile_reset(bhip_ctx* ctx) {
	if (!ctx) return BHIP_ERR_INVALID;
	(void)hipStreamSynchronize(ctx->stream);
	for (auto& r : ctx->profRecords) { ctx->eventPool.push_back(r.start); ctx->eventPool.push_back(r.stop); }
	ctx->profRecords.clear();
	return BHIP_OK;
}

// Writes one line per kernel tag: "tag launches total_ms alg_bytes alg_flops\n".  Returns the number of bytes needed (including NUL).
int bhip_profile_report(bhip_ctx* ctx, char* out, int cap) {
	if (!ctx) return BHIP_ERR_INVALID;
	(void)hipStreamSynchronize(ctx->stream);
	std::map<std::string, ProfAgg> agg;
	std::vector<std::string> order;
	for (auto& r : ctx->profRecords) {
		float ms = 0;
		if (hipEventElapsedTime(&ms, r.start, r.stop) != hipSuccess) ms = 0;
		if (!agg.count(r.tag)) order.push_back(r.tag);
		ProfAgg& a = agg[r.tag];
		a.ms += ms; a.bytes += r.algBytes; a.flops += r.algFlops; a.launches++;
	}
	std::string s;
	char line[256];
	for (auto& t : order) {
		const ProfAgg& a = agg[t];
		snprintf(line, sizeof(line), "%s %lld %.6f %.0f %.0f\n", t.c_str(), a.launches, a.ms, a.bytes, a.flops);
		s += line;
	}
	if (out && cap > 0) {
		const size_t n = std::min((size_t)cap - 1, s.size());
		memcpy(out, s.data(), n);
		out[n] = 0;
	}
	return (int)s.size() + 1;
}

}  // extern "C"
