// variants_pca_driver.cpp -- compiled host of the PCoA engine: the VariantsPcaDriver command line
// (reference src/main/scala/com/google/cloud/genomics/spark/examples/VariantsPca.scala:36-52 and
// GenomicsConf.scala:31-101) over the C ABI of include/pcoa.h.  No Python, no torch, no Spark.
//
//   main (:38-50)            conf -> data -> filterDataset -> getCallsRdd -> getSimilarityMatrix ->
//                            computePca -> emitResult -> reportIoStats -> stop
//   getVariantKey (:62-78)   Guava murmur3_128 over contig, start, end, ref, alt
//   filterDataset (:96-108)  --min-allele-frequency on INFO/AF
//   joinDatasets (:115-128)  two variant sets: inner join on the key
//   mergeDatasets (:136-148) three or more: group by key, keep complete groups
//   getCallsRdd (:153-168)   carriers per variant; variants without a varying call dropped
//   emitResult (:233-246)    name \t dataset \t pc1 \t pc2 sorted by name (+ <output-path>-pca.tsv)
//
// The data source is local: one VCF (plain or .gz via `gzip -dc`) per variant set -- the Google
// Genomics API the reference streamed from (VariantsRDD.scala) has been shut down.  Callset index =
// position in the concatenated sample lists (VariantsCommon.scala:44-45), callset id =
// "<file stem>-<i>", so `dataset` = id up to the first '-' (:235) is the file stem.
#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "pcoa.h"

namespace {

struct Conf {  // PcaConf / GenomicsConf (GenomicsConf.scala:31-101), same flag names and defaults
  std::vector<std::string> input_path;
  std::string output_path;
  std::vector<std::string> references{"chr17:41196311:41277499"};
  std::vector<std::string> variant_set_id{"3049512673186936334"};
  bool all_references = false, debug_datasets = false, has_maf = false;
  float min_allele_frequency = 0.f;
  int num_pc = 2, num_reduce_partitions = 10, gpu = 0;
  long bases_per_partition = 1000000;
  std::string client_secrets, spark_master;
};

[[noreturn]] void die(const std::string& m) {
  std::cerr << "VariantsPcaDriver: " << m << std::endl;
  std::exit(2);
}

Conf parse(int argc, char** argv) {
  Conf c;
  auto list = [&](int& i, std::vector<std::string>& dst) {
    dst.clear();
    while (i + 1 < argc && std::strncmp(argv[i + 1], "--", 2) != 0) dst.push_back(argv[++i]);
  };
  auto one = [&](int& i) -> std::string {
    if (i + 1 >= argc) die(std::string("missing value for ") + argv[i]);
    return argv[++i];
  };
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--input-path") list(i, c.input_path);
    else if (a == "--references") list(i, c.references);
    else if (a == "--variant-set-id") list(i, c.variant_set_id);
    else if (a == "--output-path") c.output_path = one(i);
    else if (a == "--num-pc") c.num_pc = std::atoi(one(i).c_str());
    else if (a == "--min-allele-frequency") { c.min_allele_frequency = std::strtof(one(i).c_str(), nullptr); c.has_maf = true; }
    else if (a == "--all-references") c.all_references = true;
    else if (a == "--debug-datasets") c.debug_datasets = true;
    else if (a == "--bases-per-partition") c.bases_per_partition = std::atol(one(i).c_str());
    else if (a == "--num-reduce-partitions") c.num_reduce_partitions = std::atoi(one(i).c_str());
    else if (a == "--client-secrets") c.client_secrets = one(i);
    else if (a == "--spark-master") c.spark_master = one(i);
    else if (a == "--gpu") c.gpu = std::atoi(one(i).c_str());
    else die("unknown flag " + a);
  }
  return c;
}

// ---- Double.toString (what Scala's string interpolation prints, :239) --------------------------
template <typename T>  // T = double: Double.toString; T = float: Float.toString (same layout rules)
std::string java_number(T d) {
  if (std::isnan(d)) return "NaN";
  if (std::isinf(d)) return d > 0 ? "Infinity" : "-Infinity";
  if (d == 0) return std::signbit(d) ? "-0.0" : "0.0";
  char buf[64];
  auto r = std::to_chars(buf, buf + sizeof(buf), std::fabs(d), std::chars_format::scientific);  // shortest round trip
  std::string s(buf, r.ptr);                      // d.ddddde[+-]XX
  const size_t epos = s.find('e');
  std::string mant = s.substr(0, epos);
  int e10 = std::atoi(s.c_str() + epos + 1);
  std::string digits;
  for (char ch : mant)
    if (ch != '.') digits.push_back(ch);
  while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
  const std::string sign = d < 0 ? "-" : "";
  const double ad = std::fabs(d);
  if (ad >= 1e-3 && ad < 1e7) {
    std::string whole, frac;
    if (e10 >= 0) {
      whole = digits.substr(0, std::min<size_t>(digits.size(), (size_t)e10 + 1));
      while ((int)whole.size() < e10 + 1) whole.push_back('0');
      frac = digits.size() > (size_t)e10 + 1 ? digits.substr((size_t)e10 + 1) : "0";
    } else {
      whole = "0";
      frac = std::string((size_t)(-e10 - 1), '0') + digits;
    }
    return sign + whole + "." + frac;
  }
  return sign + digits.substr(0, 1) + "." + (digits.size() > 1 ? digits.substr(1) : "0") + "E" + std::to_string(e10);
}
std::string java_double(double d) { return java_number<double>(d); }

// ---- Guava Hashing.murmur3_128().hashBytes(..).toString() -------------------------------------
inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t fmix64(uint64_t k) {
  k ^= k >> 33; k *= 0xFF51AFD7ED558CCDULL; k ^= k >> 33; k *= 0xC4CEB9FE1A85EC53ULL; k ^= k >> 33;
  return k;
}
std::string murmur3_128_hex(const std::string& data) {
  const uint64_t c1 = 0x87C37B91114253D5ULL, c2 = 0x4CF5AD432745937FULL;
  uint64_t h1 = 0, h2 = 0;
  const size_t n = data.size(), nblocks = n / 16;
  auto rd = [&](size_t off, size_t len) { uint64_t v = 0; std::memcpy(&v, data.data() + off, len); return v; };
  for (size_t i = 0; i < nblocks; ++i) {
    uint64_t k1 = rd(16 * i, 8), k2 = rd(16 * i + 8, 8);
    k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52DCE729;
    k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
    h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495AB5;
  }
  const size_t t = n - 16 * nblocks;
  if (t > 8) { uint64_t k2 = rd(16 * nblocks + 8, t - 8); k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; }
  if (t > 0) { uint64_t k1 = rd(16 * nblocks, std::min<size_t>(t, 8)); k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1; }
  h1 ^= n; h2 ^= n; h1 += h2; h2 += h1; h1 = fmix64(h1); h2 = fmix64(h2); h1 += h2; h2 += h1;
  char out[33];
  unsigned char b[16];
  std::memcpy(b, &h1, 8); std::memcpy(b + 8, &h2, 8);  // little-endian host
  for (int i = 0; i < 16; ++i) std::snprintf(out + 2 * i, 3, "%02x", b[i]);
  return std::string(out, 32);
}

// ---- variants -------------------------------------------------------------------------------
struct CallData { bool has_variation; int32_t callset; };  // case class CallData (:288)
struct Variant {
  std::string key;                 // getVariantKey
  bool has_af = false; float af = 0.f;
  std::vector<CallData> calls;     // extractCallInfo (:56-60)
};
struct Dataset {
  std::vector<std::string> ids, names;
  std::vector<Variant> variants;
};

// contig normalisation of the reference (VariantsRDD.scala:103-110): [a-z]*[0-9]+ -> the number
bool normalize_contig(const std::string& name, std::string& out) {
  size_t i = 0;
  while (i < name.size() && std::isalpha((unsigned char)name[i])) ++i;
  if (i == name.size()) return false;
  for (size_t j = i; j < name.size(); ++j)
    if (!std::isdigit((unsigned char)name[j])) return false;
  out = name.substr(i);
  return true;
}

struct Region { std::string contig; long start, end; };
std::vector<Region> parse_references(const std::string& spec) {
  std::vector<Region> out;
  std::stringstream ss(spec);
  std::string tup;
  while (std::getline(ss, tup, ',')) {
    if (tup.empty()) continue;
    const size_t a = tup.find(':'), b = tup.rfind(':');
    if (a == std::string::npos || a == b) die("bad --references tuple " + tup);
    Region r;
    std::string ctg = tup.substr(0, a);
    if (!normalize_contig(ctg, r.contig)) r.contig = ctg;
    r.start = std::atol(tup.substr(a + 1, b - a - 1).c_str());
    r.end = std::atol(tup.substr(b + 1).c_str());
    out.push_back(r);
  }
  return out;
}

std::vector<std::string> split(const std::string& s, char sep) {
  std::vector<std::string> out;
  size_t b = 0;
  for (;;) {
    const size_t e = s.find(sep, b);
    out.push_back(s.substr(b, e == std::string::npos ? std::string::npos : e - b));
    if (e == std::string::npos) break;
    b = e + 1;
  }
  return out;
}

struct LineReader {  // plain file or `gzip -dc` pipe
  FILE* f = nullptr;
  bool piped = false;
  explicit LineReader(const std::string& path) {
    if (path.size() > 3 && path.substr(path.size() - 3) == ".gz") {
      std::string cmd = "gzip -dc '" + path + "'";
      f = popen(cmd.c_str(), "r");
      piped = true;
    } else {
      f = std::fopen(path.c_str(), "r");
    }
    if (!f) die("cannot open " + path);
  }
  ~LineReader() { if (f) { if (piped) pclose(f); else std::fclose(f); } }
  bool next(std::string& line) {
    line.clear();
    char buf[1 << 16];
    while (std::fgets(buf, sizeof(buf), f)) {
      line += buf;
      if (!line.empty() && line.back() == '\n') { line.pop_back(); return true; }
    }
    return !line.empty();
  }
};

Dataset load_vcf(const std::string& path, const std::vector<Region>& regions, int32_t index_base, bool debug) {
  Dataset d;
  std::string stem = path.substr(path.find_last_of('/') == std::string::npos ? 0 : path.find_last_of('/') + 1);
  stem = stem.substr(0, stem.find('.'));
  std::replace(stem.begin(), stem.end(), '-', '_');
  LineReader in(path);
  std::string line;
  bool header = false;
  while (in.next(line)) {
    if (line.rfind("##", 0) == 0) continue;
    if (line.rfind("#CHROM", 0) == 0) {
      auto cols = split(line, '\t');
      for (size_t i = 9; i < cols.size(); ++i) {
        d.names.push_back(cols[i]);
        d.ids.push_back(stem + "-" + std::to_string(i - 9));
      }
      header = true;
      continue;
    }
    if (!header) die("VCF header line (#CHROM) missing in " + path);
    auto rec = split(line, '\t');
    if (rec.size() < 10) continue;
    std::string contig;
    if (!normalize_contig(rec[0], contig)) continue;  // X, Y, MT ... dropped as in the reference
    const long start = std::atol(rec[1].c_str()) - 1;
    if (!regions.empty()) {
      bool in_region = false;
      for (const auto& r : regions) in_region = in_region || (r.contig == contig && r.start <= start && start < r.end);
      if (!in_region) continue;
    }
    auto fmt = split(rec[8], ':');
    int gti = -1;
    for (size_t i = 0; i < fmt.size(); ++i) if (fmt[i] == "GT") gti = (int)i;
    if (gti < 0) continue;
    Variant v;
    std::string alt;
    for (const auto& a : split(rec[4], ',')) if (a != ".") alt += a;
    const long end = start + (long)rec[3].size();
    if (debug) std::printf("%s: (%ld, %ld) ref=%s alt=%s\n", contig.c_str(), start, end, rec[3].c_str(), alt.c_str());
    std::string buf = contig;
    int64_t s64 = start, e64 = end;
    buf.append(reinterpret_cast<const char*>(&s64), 8);
    buf.append(reinterpret_cast<const char*>(&e64), 8);
    buf += rec[3];
    buf += alt;
    v.key = murmur3_128_hex(buf);
    for (const auto& item : split(rec[7], ';')) {
      if (item.rfind("AF=", 0) == 0) { v.af = std::strtof(item.c_str() + 3, nullptr); v.has_af = true; }
    }
    for (size_t i = 9; i < rec.size() && i - 9 < d.ids.size(); ++i) {
      auto parts = split(rec[i], ':');
      const std::string gt = gti < (int)parts.size() ? parts[(size_t)gti] : ".";
      bool has_variation = false;  // genotype.foldLeft(false)(_ || _ > 0), :58
      size_t b = 0;
      for (size_t p = 0; p <= gt.size(); ++p) {
        if (p == gt.size() || gt[p] == '/' || gt[p] == '|') {
          const std::string allele = gt.substr(b, p - b);
          if (!allele.empty() && allele != "." && std::atoi(allele.c_str()) > 0) has_variation = true;
          b = p + 1;
        }
      }
      v.calls.push_back({has_variation, index_base + (int32_t)(i - 9)});
    }
    d.variants.push_back(std::move(v));
  }
  if (!header) die("no #CHROM header in " + path);
  return d;
}

void check(pcoa_ctx* ctx, int rc, const char* what) {
  if (rc != PCOA_OK) die(std::string(what) + ": " + pcoa_last_error(ctx));
}

}  // namespace

int main(int argc, char** argv) {
  Conf conf = parse(argc, argv);
  if (conf.input_path.empty())
    die("--input-path <file.vcf[.gz]> [more files] is required: the Google Genomics API the reference read "
        "from has been shut down");
  // VariantsCommon (VariantsCommon.scala:33-66): callset index/name maps + one dataset per variant set
  std::vector<Dataset> data;
  std::vector<std::string> ids, names;
  if (conf.input_path.size() > 1) std::printf("Running PCA on %zu datasets.\n", conf.input_path.size());
  for (size_t k = 0; k < conf.input_path.size(); ++k) {
    std::vector<Region> regions;
    if (!conf.all_references && !conf.references.empty())
      regions = parse_references(conf.references[std::min(k, conf.references.size() - 1)]);
    data.push_back(load_vcf(conf.input_path[k], regions, (int32_t)ids.size(), conf.debug_datasets));
    ids.insert(ids.end(), data.back().ids.begin(), data.back().ids.end());
    names.insert(names.end(), data.back().names.begin(), data.back().names.end());
  }
  const int32_t n = (int32_t)ids.size();
  std::printf("Matrix size: %d.\n", n);
  if (n == 0) die("no samples");

  // filterDataset (:96-108)
  if (conf.has_maf) {
    for (auto& d : data) {
      std::printf("Min allele frequency %s.\n", java_number<float>(conf.min_allele_frequency).c_str());  // per dataset (:43)
      std::vector<Variant> kept;
      for (auto& v : d.variants)
        if (v.has_af && v.af >= conf.min_allele_frequency) kept.push_back(std::move(v));
      d.variants.swap(kept);
    }
  }

  // getCallsRdd (:153-168)
  std::vector<std::vector<CallData>> callsets;
  if (data.size() == 1) {
    for (auto& v : data[0].variants) callsets.push_back(std::move(v.calls));
  } else if (data.size() == 2) {  // joinDatasets
    std::unordered_map<std::string, std::vector<const Variant*>> right;
    for (const auto& v : data[1].variants) right[v.key].push_back(&v);
    for (const auto& v : data[0].variants) {
      auto it = right.find(v.key);
      if (it == right.end()) continue;
      for (const Variant* w : it->second) {
        std::vector<CallData> joined = v.calls;
        joined.insert(joined.end(), w->calls.begin(), w->calls.end());
        callsets.push_back(std::move(joined));
      }
    }
  } else {  // mergeDatasets
    std::map<std::string, std::vector<const Variant*>> groups;
    for (const auto& d : data)
      for (const auto& v : d.variants) groups[v.key].push_back(&v);
    for (const auto& g : groups) {
      if (g.second.size() != data.size()) continue;
      std::vector<CallData> merged;
      for (const Variant* v : g.second) merged.insert(merged.end(), v->calls.begin(), v->calls.end());
      callsets.push_back(std::move(merged));
    }
  }
  std::vector<int32_t> sample_idx;
  std::vector<int64_t> row_offsets{0};
  for (const auto& calls : callsets) {
    size_t before = sample_idx.size();
    for (const auto& c : calls)
      if (c.has_variation) sample_idx.push_back(c.callset);
    if (sample_idx.size() > before) row_offsets.push_back((int64_t)sample_idx.size());  // drop empty (:166)
  }
  if (sample_idx.empty()) sample_idx.push_back(0);

  // getSimilarityMatrix (:182-191) and computePca (:198-231) on the GPU
  pcoa_ctx* ctx = nullptr;
  if (pcoa_create(&ctx, n, conf.gpu, PCOA_FLAG_DEFAULT) != PCOA_OK) die(std::string("pcoa_create: ") + pcoa_last_error(nullptr));
  check(ctx, pcoa_accumulate_calls(ctx, sample_idx.data(), row_offsets.data(), (int64_t)row_offsets.size() - 1),
        "getSimilarityMatrix");
  check(ctx, pcoa_gram_finalize(ctx), "getSimilarityMatrix");
  if (conf.num_pc < 2)  // the reference reads array(i + pca.numRows) unconditionally (:230)
    die("computePca emits exactly PC1 and PC2 (VariantsPca.scala:229-230); --num-pc must be >= 2");
  std::vector<double> comps((size_t)conf.num_pc * (size_t)n), lam((size_t)conf.num_pc);
  int32_t nonzero = 0;
  check(ctx, pcoa_compute(ctx, conf.num_pc, comps.data(), lam.data(), &nonzero), "computePca");
  std::printf("Non zero rows in matrix: %d / %d.\n", nonzero, n);

  // emitResult (:233-246)
  struct Row { std::string name, dataset; double pc1, pc2; };
  std::vector<Row> rows;
  for (int32_t i = 0; i < n; ++i)
    rows.push_back({names[(size_t)i], ids[(size_t)i].substr(0, ids[(size_t)i].find('-')), comps[(size_t)i],
                    comps[(size_t)i + (size_t)n]});
  std::stable_sort(rows.begin(), rows.end(), [](const Row& a, const Row& b) { return a.name < b.name; });
  for (const auto& r : rows)
    std::printf("%s\t%s\t%s\t%s\n", r.name.c_str(), r.dataset.c_str(), java_double(r.pc1).c_str(), java_double(r.pc2).c_str());
  if (!conf.output_path.empty()) {
    std::ofstream out(conf.output_path + "-pca.tsv");
    for (const auto& r : rows)
      out << r.name << "\t" << java_double(r.pc1) << "\t" << java_double(r.pc2) << "\t" << r.dataset << "\n";
  }

  // reportIoStats (:48) / stop (:49)
  pcoa_timings t;
  if (pcoa_get_timings(ctx, &t) == PCOA_OK)
    std::fprintf(stderr, "Variants accumulated: %lld; Gram kernel %.3f ms; PCoA %.3f ms\n", (long long)t.gram_variants,
                 1e3 * t.gram_kernel_seconds, 1e3 * t.compute_total_seconds);
  pcoa_destroy(ctx);
  return 0;
}
