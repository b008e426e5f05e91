// scidac_io.cpp -- SciDAC/LIME gauge configuration files.  Host code, no GPU.
//
// Replaces, for gauge fields, what QEX's loadGauge / saveGauge do (src/gauge/gaugeUtils.nim:87-122)
// through Reader / Writer (src/io/readerQiolite.nim:37-239, src/io/writerQiolite.nim:28-187).  Those sit
// on the Nim package `qiolite` (`scidacio`, qex.nimble requires; not vendored in the reference tree), which
// implements the SciDAC file format of USQCD's QIO over c-lime.  The format is restated here from those
// published specifications:
//   LIME record = 144-byte header {magic 0x456789ab, version 1, flags (bit15 MB, bit14 ME), 64-bit data
//     length, 128-byte NUL-padded type} + data, padded to a multiple of 8 bytes; all big-endian.
//   message 1: "scidac-private-file-xml" (<scidacFile> version, spacetime, dims, volfmt) , "scidac-file-xml"
//   message 2: "scidac-private-record-xml" (<scidacRecord> version, date, recordtype, datatype, precision,
//     colors, typesize, datacount), "scidac-record-xml", "scidac-binary-data" | "ildg-binary-data",
//     "scidac-checksum" (<scidacChecksum> version, suma, sumb in hex)
//   binary data: sites in lexicographic order (x fastest), per site `datacount` objects (the 4 directions)
//     of `typesize` bytes = 3x3 complex row-major (re, im), IEEE big-endian, F or D.
//   checksum: crc = crc32(site bytes as stored); suma ^= rotl32(crc, rank % 29); sumb ^= rotl32(crc, rank % 31),
//     rank = lexicographic site index (QIO's DML_checksum_accum; CRC-32 as in src/io/crc32.nim).
// What the reference itself fixes: the record's datatype names "QDP_F3_ColorMatrix" / "QDP_D3_ColorMatrix"
// (src/io/qioInternal.nim:43-51), precision "F"/"D", colors, typesize, datacount (writerQiolite.nim:124-137),
// the default metadata strings (gaugeUtils.nim:108-109) and x-fastest site order (hyperindex, :63-66).
// Host field format as everywhere in this library: V=1 even-odd, double g[vol][4][3][3][2].
#include "../../include/qexhip.h"
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <sys/stat.h>
#include <ctime>
#include <string>
#include <vector>

void qexhip_set_error(const char *fmt, ...);

namespace {
// CRC-32 as src/io/crc32.nim: reflected polynomial 0xedb88320 (:5), table of the 256 byte remainders (:8-16), start
// 0xffffffff, final complement (:25-37).  The reference's own known answer (crc32.nim:103-106) is held in tests/test_scidac_io.py.
struct CrcTable {
  uint32_t t[256];
  CrcTable() {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t c = i;
      for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      t[i] = c;
    }
  }
};
uint32_t crc32(const unsigned char *p, size_t n) {
  static const CrcTable tab;                 // thread-safe one-time initialisation
  const uint32_t *crc_table = tab.t;
  uint32_t c = 0xFFFFFFFFu;
  for (size_t i = 0; i < n; i++) c = crc_table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}
inline uint32_t rotl32(uint32_t v, unsigned r) { return r ? (v << r) | (v >> (32 - r)) : v; }

struct Checksum {
  uint32_t a = 0, b = 0;
  void add(const unsigned char *site, size_t n, uint64_t rank) {
    const uint32_t c = crc32(site, n);
    a ^= rotl32(c, (unsigned)(rank % 29));
    b ^= rotl32(c, (unsigned)(rank % 31));
  }
};

void put_be64(unsigned char *p, uint64_t v) { for (int i = 0; i < 8; i++) p[i] = (unsigned char)(v >> (56 - 8 * i)); }
uint64_t get_be64(const unsigned char *p) { uint64_t v = 0; for (int i = 0; i < 8; i++) v = (v << 8) | p[i]; return v; }

struct LimeHeader {
  bool mb, me;
  uint64_t len;
  std::string type;
};
bool write_record(FILE *f, const char *type, bool mb, bool me, const void *data, uint64_t len) {
  unsigned char h[144];
  memset(h, 0, sizeof h);
  h[0] = 0x45; h[1] = 0x67; h[2] = 0x89; h[3] = 0xab;
  h[4] = 0; h[5] = 1;
  h[6] = (unsigned char)((mb ? 0x80 : 0) | (me ? 0x40 : 0));
  put_be64(h + 8, len);
  strncpy((char *)h + 16, type, 127);
  if (fwrite(h, 1, 144, f) != 144) return false;
  if (len && fwrite(data, 1, len, f) != len) return false;
  static const unsigned char zero[8] = {0};
  const size_t pad = (8 - len % 8) % 8;
  return !pad || fwrite(zero, 1, pad, f) == pad;
}
// 1: header read, 0: clean end of file, -1: error
int read_header(FILE *f, LimeHeader &h) {
  unsigned char b[144];
  size_t n = fread(b, 1, 144, f);
  if (n == 0) return 0;
  if (n != 144 || b[0] != 0x45 || b[1] != 0x67 || b[2] != 0x89 || b[3] != 0xab) return -1;
  h.mb = b[6] & 0x80; h.me = b[6] & 0x40;
  h.len = get_be64(b + 8);
  b[143] = 0;
  h.type = (const char *)b + 16;
  return 1;
}
// bytes left between the read position and the end of the file (lengths in LIME headers are untrusted input)
// (the size comes from fstat: no seek to the end and back per record, which would drop the stdio read-ahead each time)
uint64_t bytes_left(FILE *f) {
  const off_t here = ftello(f);
  struct stat st;
  if (here < 0 || fstat(fileno(f), &st) != 0) return 0;
  return st.st_size > here ? (uint64_t)(st.st_size - here) : 0;
}
bool skip_data(FILE *f, uint64_t len) {
  if (len > bytes_left(f)) { qexhip_set_error("lime: record of %llu bytes runs past the end of the file", (unsigned long long)len); return false; }
  const uint64_t padded = len + (8 - len % 8) % 8;
  return fseeko(f, (off_t)std::min<uint64_t>(padded, bytes_left(f)), SEEK_CUR) == 0;
}
bool read_text(FILE *f, uint64_t len, std::string &s) {
  // The reference's reader puts no cap on user file / record XML (readerQiolite.nim), and qexhip_io_metadata is there to
  // return it: the bound is what the file still holds (a hostile length cannot exceed that), plus 256 MiB as a sanity limit.
  if (len > (256u << 20) || len > bytes_left(f)) {
    qexhip_set_error("lime: text record of %llu bytes (limit 256 MiB, and no more than the file holds)", (unsigned long long)len);
    return false;
  }
  s.resize(len);
  if (len && fread(&s[0], 1, len, f) != len) return false;
  while (!s.empty() && s.back() == '\0') s.pop_back();
  return fseeko(f, (off_t)((8 - len % 8) % 8), SEEK_CUR) == 0;
}
std::string xml_tag(const std::string &x, const char *tag) {
  const std::string o = std::string("<") + tag + ">", c = std::string("</") + tag + ">";
  size_t i = x.find(o);
  if (i == std::string::npos) return "";
  i += o.size();
  size_t j = x.find(c, i);
  return j == std::string::npos ? "" : x.substr(i, j - i);
}
inline size_t eo_index(const int lat[4], const int x[4]) {
  const size_t lex = x[0] + (size_t)lat[0] * (x[1] + (size_t)lat[1] * (x[2] + (size_t)lat[2] * x[3]));
  const size_t vol = (size_t)lat[0] * lat[1] * lat[2] * lat[3];
  return lex / 2 + (((x[0] + x[1] + x[2] + x[3]) & 1) ? vol / 2 : 0);
}
template <class T> void to_be(unsigned char *dst, T v) {
  unsigned char b[sizeof(T)];
  memcpy(b, &v, sizeof(T));
  const uint16_t one = 1;
  const bool little = *(const unsigned char *)&one == 1;
  for (size_t i = 0; i < sizeof(T); i++) dst[i] = little ? b[sizeof(T) - 1 - i] : b[i];
}
template <class T> T from_be(const unsigned char *src) {
  unsigned char b[sizeof(T)];
  const uint16_t one = 1;
  const bool little = *(const unsigned char *)&one == 1;
  for (size_t i = 0; i < sizeof(T); i++) b[i] = little ? src[sizeof(T) - 1 - i] : src[i];
  T v;
  memcpy(&v, b, sizeof(T));
  return v;
}

struct FileInfo {
  int lat[4] = {0, 0, 0, 0};
  int nd = 0;
  char prec = 0;
  int colors = 0, typesize = 0, datacount = 0;
  std::string datatype, file_md, record_md, date;
  off_t data_off = -1;
  uint64_t data_len = 0;
  bool have_sum = false;
  uint32_t suma = 0, sumb = 0;
};
int scan(FILE *f, FileInfo &I) {
  LimeHeader h;
  int r;
  while ((r = read_header(f, h)) == 1) {
    std::string t;
    if (h.type == "scidac-private-file-xml") {
      if (!read_text(f, h.len, t)) return -1;
      I.nd = atoi(xml_tag(t, "spacetime").c_str());
      std::string d = xml_tag(t, "dims");
      if (I.nd != 4 || sscanf(d.c_str(), "%d %d %d %d", &I.lat[0], &I.lat[1], &I.lat[2], &I.lat[3]) != 4) {
        qexhip_set_error("scidac: only 4-dimensional lattices are supported");
        return -1;
      }
    } else if (h.type == "scidac-file-xml") {
      if (!read_text(f, h.len, I.file_md)) return -1;
    } else if (h.type == "scidac-private-record-xml") {
      if (I.data_off >= 0) break;                        // a second field record: stop at the first
      if (!read_text(f, h.len, t)) return -1;
      I.datatype = xml_tag(t, "datatype");
      const std::string p = xml_tag(t, "precision");
      I.prec = p.empty() ? 0 : p[0];
      I.colors = atoi(xml_tag(t, "colors").c_str());
      I.typesize = atoi(xml_tag(t, "typesize").c_str());
      I.datacount = atoi(xml_tag(t, "datacount").c_str());
      I.date = xml_tag(t, "date");
    } else if (h.type == "scidac-record-xml") {
      if (!read_text(f, h.len, I.record_md)) return -1;
    } else if (h.type == "scidac-binary-data" || h.type == "ildg-binary-data") {
      I.data_off = ftello(f);
      I.data_len = h.len;
      if (!skip_data(f, h.len)) return -1;
    } else if (h.type == "scidac-checksum") {
      if (!read_text(f, h.len, t)) return -1;
      I.suma = (uint32_t)strtoul(xml_tag(t, "suma").c_str(), nullptr, 16);
      I.sumb = (uint32_t)strtoul(xml_tag(t, "sumb").c_str(), nullptr, 16);
      I.have_sum = true;
      if (I.data_off >= 0) break;
    } else if (!skip_data(f, h.len)) {
      return -1;
    }
  }
  if (r < 0) { qexhip_set_error("scidac: not a LIME file (bad record header)"); return -1; }
  if (I.nd != 4 || I.data_off < 0) { qexhip_set_error("scidac: no lattice dimensions or no binary data record found"); return -1; }
  return 0;
}
}  // namespace

static int io_gauge_info_impl(const char *path, int lat[4], char *precision, int *checksums_present) {
  if (!path) return QEXHIP_ERR_ARG;
  FILE *f = fopen(path, "rb");
  if (!f) { qexhip_set_error("scidac: cannot open file"); return QEXHIP_ERR_ARG; }
  FileInfo I;
  const int rc = scan(f, I);
  fclose(f);
  if (rc) return QEXHIP_ERR_ARG;
  if (lat) for (int i = 0; i < 4; i++) lat[i] = I.lat[i];
  if (precision) *precision = I.prec;
  if (checksums_present) *checksums_present = I.have_sum;
  return 0;
}

static int read_gauge_impl(const char *path, const int lat[4], int t0, int nt, double *g, unsigned *suma, unsigned *sumb);
static int io_read_gauge_impl(const char *path, const int lat[4], double *g, unsigned *suma, unsigned *sumb) {
  if (!path || !lat || !g) return QEXHIP_ERR_ARG;
  return read_gauge_impl(path, lat, 0, lat[3], g, suma, sumb);
}
// one rank's slab t0 <= t < t0 + nt of a file holding the GLOBAL lattice `lat`; g is the local field (its own
// even-odd order).  The whole record is read, so the checksums are still verified.
static int io_read_gauge_slab_impl(const char *path, const int lat[4], int t0, int nt, double *g) {
  if (!path || !lat || !g || t0 < 0 || nt < 2 || (nt & 1) || (t0 & 1) || t0 + nt > lat[3]) return QEXHIP_ERR_ARG;
  return read_gauge_impl(path, lat, t0, nt, g, nullptr, nullptr);
}
static int read_gauge_impl(const char *path, const int lat[4], int t0, int nt, double *g, unsigned *suma, unsigned *sumb) {
  FILE *f = fopen(path, "rb");
  if (!f) { qexhip_set_error("scidac: cannot open file"); return QEXHIP_ERR_ARG; }
  FileInfo I;
  if (scan(f, I)) { fclose(f); return QEXHIP_ERR_ARG; }
  for (int i = 0; i < 4; i++)
    if (I.lat[i] != lat[i]) { fclose(f); qexhip_set_error("scidac: file lattice differs from the requested one"); return QEXHIP_ERR_ARG; }
  const size_t vol = (size_t)lat[0] * lat[1] * lat[2] * lat[3];
  // precision from the record; files without a private record (plain ILDG) are sized by their length
  int wsz = I.prec == 'F' ? 4 : (I.prec == 'D' ? 8 : 0);
  if (!wsz) wsz = I.data_len == vol * 72 * 4 ? 4 : 8;
  const size_t site_bytes = (size_t)72 * wsz;
  if (I.data_len != vol * site_bytes || (I.typesize && (I.typesize != 18 * wsz || I.datacount != 4 || I.colors != 3))) {
    fclose(f);
    qexhip_set_error("scidac: record is not a 4 x SU(3) gauge field of this lattice");
    return QEXHIP_ERR_ARG;
  }
  fseeko(f, I.data_off, SEEK_SET);
  std::vector<unsigned char> buf(site_bytes * lat[0]);
  Checksum cs;
  uint64_t rank = 0;
  int x[4];
  for (x[3] = 0; x[3] < lat[3]; x[3]++)
    for (x[2] = 0; x[2] < lat[2]; x[2]++)
      for (x[1] = 0; x[1] < lat[1]; x[1]++) {
        if (fread(buf.data(), 1, buf.size(), f) != buf.size()) { fclose(f); qexhip_set_error("scidac: short read"); return QEXHIP_ERR_ARG; }
        for (x[0] = 0; x[0] < lat[0]; x[0]++, rank++) {
          const unsigned char *s = buf.data() + site_bytes * x[0];
          cs.add(s, site_bytes, rank);
          if (x[3] < t0 || x[3] >= t0 + nt) continue;
          const int ll[4] = {lat[0], lat[1], lat[2], nt}, xl[4] = {x[0], x[1], x[2], x[3] - t0};
          double *d = g + eo_index(ll, xl) * 72;
          if (wsz == 8) for (int k = 0; k < 72; k++) d[k] = from_be<double>(s + 8 * k);
          else for (int k = 0; k < 72; k++) d[k] = (double)from_be<float>(s + 4 * k);
        }
      }
  fclose(f);
  if (suma) *suma = cs.a;
  if (sumb) *sumb = cs.b;
  if (I.have_sum && (cs.a != I.suma || cs.b != I.sumb)) {
    qexhip_set_error("scidac: checksum mismatch");
    return QEXHIP_ERR_IO;
  }
  return 0;
}

static int io_write_gauge_impl(const char *path, const int lat[4], const double *g, char precision,
                                     const char *file_md, const char *record_md) {
  if (!path || !lat || !g || (precision != 'F' && precision != 'D')) return QEXHIP_ERR_ARG;
  if (!file_md) file_md = "<?xml version=\"1.0\"?>\n<note>generated by QEX</note>\n";            // gaugeUtils.nim:108
  if (!record_md) record_md = "<?xml version=\"1.0\"?>\n<note>gauge configuration</note>\n";      // gaugeUtils.nim:109
  FILE *f = fopen(path, "wb");
  if (!f) { qexhip_set_error("scidac: cannot create file"); return QEXHIP_ERR_ARG; }
  const int wsz = precision == 'F' ? 4 : 8;
  const size_t vol = (size_t)lat[0] * lat[1] * lat[2] * lat[3];
  const size_t site_bytes = (size_t)72 * wsz;
  char xml[1024];
  bool ok = true;
  snprintf(xml, sizeof xml,
           "<?xml version=\"1.0\" encoding=\"UTF-8\"?><scidacFile><version>1.1</version><spacetime>4</spacetime>"
           "<dims>%d %d %d %d </dims><volfmt>0</volfmt></scidacFile>", lat[0], lat[1], lat[2], lat[3]);
  ok = ok && write_record(f, "scidac-private-file-xml", true, false, xml, strlen(xml) + 1);
  ok = ok && write_record(f, "scidac-file-xml", false, true, file_md, strlen(file_md) + 1);
  char date[64];
  time_t now = time(nullptr);
  strftime(date, sizeof date, "%a %b %d %H:%M:%S %Y UTC", gmtime(&now));
  snprintf(xml, sizeof xml,
           "<?xml version=\"1.0\" encoding=\"UTF-8\"?><scidacRecord><version>1.1</version><date>%s</date>"
           "<recordtype>0</recordtype><datatype>QDP_%c3_ColorMatrix</datatype><precision>%c</precision>"
           "<colors>3</colors><typesize>%d</typesize><datacount>4</datacount></scidacRecord>",
           date, precision, precision, 18 * wsz);
  ok = ok && write_record(f, "scidac-private-record-xml", true, false, xml, strlen(xml) + 1);
  ok = ok && write_record(f, "scidac-record-xml", false, false, record_md, strlen(record_md) + 1);
  // binary record: header first, then the sites streamed row by row
  {
    unsigned char h[144];
    memset(h, 0, sizeof h);
    h[0] = 0x45; h[1] = 0x67; h[2] = 0x89; h[3] = 0xab; h[5] = 1;
    put_be64(h + 8, (uint64_t)vol * site_bytes);
    strncpy((char *)h + 16, "scidac-binary-data", 127);
    ok = ok && fwrite(h, 1, 144, f) == 144;
  }
  std::vector<unsigned char> buf(site_bytes * lat[0]);
  Checksum cs;
  uint64_t rank = 0;
  int x[4];
  for (x[3] = 0; x[3] < lat[3] && ok; x[3]++)
    for (x[2] = 0; x[2] < lat[2] && ok; x[2]++)
      for (x[1] = 0; x[1] < lat[1] && ok; x[1]++) {
        for (x[0] = 0; x[0] < lat[0]; x[0]++, rank++) {
          unsigned char *s = buf.data() + site_bytes * x[0];
          const double *d = g + eo_index(lat, x) * 72;
          if (wsz == 8) for (int k = 0; k < 72; k++) to_be<double>(s + 8 * k, d[k]);
          else for (int k = 0; k < 72; k++) to_be<float>(s + 4 * k, (float)d[k]);
          cs.add(s, site_bytes, rank);
        }
        ok = fwrite(buf.data(), 1, buf.size(), f) == buf.size();
      }
  // (vol * site_bytes is a multiple of 8: no padding)
  snprintf(xml, sizeof xml,
           "<?xml version=\"1.0\" encoding=\"UTF-8\"?><scidacChecksum><version>1.0</version><suma>%x</suma><sumb>%x</sumb>"
           "</scidacChecksum>", cs.a, cs.b);
  ok = ok && write_record(f, "scidac-checksum", false, true, xml, strlen(xml) + 1);
  ok = (fclose(f) == 0) && ok;
  if (!ok) { qexhip_set_error("scidac: write failed"); return QEXHIP_ERR_ARG; }
  return 0;
}

// ---- generic site records: what Writer.write / Reader.read do for any field type (writerQiolite.nim:96-166,
// readerQiolite.nim:120-200): `site_bytes` per site in x-fastest order, every `word_bytes`-wide word big-endian.
// data: host array in the library's even-odd site order, words in host byte order.
static int io_write_field_impl(const char *path, const int lat[4], const void *data, int site_bytes, int word_bytes,
                                     const char *datatype, char precision, int colors, int datacount, const char *file_md,
                                     const char *record_md) {
  if (!path || !lat || !data || !datatype || site_bytes < 1 || (word_bytes != 4 && word_bytes != 8) || site_bytes % word_bytes ||
      datacount < 1 || site_bytes % datacount)
    return QEXHIP_ERR_ARG;
  if (!file_md) file_md = "<?xml version=\"1.0\"?>\n<note>generated by QEX</note>\n";
  if (!record_md) record_md = "<?xml version=\"1.0\"?>\n<note>field</note>\n";
  FILE *f = fopen(path, "wb");
  if (!f) { qexhip_set_error("scidac: cannot create file"); return QEXHIP_ERR_ARG; }
  const size_t vol = (size_t)lat[0] * lat[1] * lat[2] * lat[3];
  char xml[1024];
  bool ok = true;
  snprintf(xml, sizeof xml,
           "<?xml version=\"1.0\" encoding=\"UTF-8\"?><scidacFile><version>1.1</version><spacetime>4</spacetime>"
           "<dims>%d %d %d %d </dims><volfmt>0</volfmt></scidacFile>", lat[0], lat[1], lat[2], lat[3]);
  ok = ok && write_record(f, "scidac-private-file-xml", true, false, xml, strlen(xml) + 1);
  ok = ok && write_record(f, "scidac-file-xml", false, true, file_md, strlen(file_md) + 1);
  char date[64];
  time_t now = time(nullptr);
  strftime(date, sizeof date, "%a %b %d %H:%M:%S %Y UTC", gmtime(&now));
  snprintf(xml, sizeof xml,
           "<?xml version=\"1.0\" encoding=\"UTF-8\"?><scidacRecord><version>1.1</version><date>%s</date>"
           "<recordtype>0</recordtype><datatype>%s</datatype><precision>%c</precision>"
           "<colors>%d</colors><typesize>%d</typesize><datacount>%d</datacount></scidacRecord>",
           date, datatype, precision, colors, site_bytes / datacount, datacount);
  ok = ok && write_record(f, "scidac-private-record-xml", true, false, xml, strlen(xml) + 1);
  ok = ok && write_record(f, "scidac-record-xml", false, false, record_md, strlen(record_md) + 1);
  {
    unsigned char h[144];
    memset(h, 0, sizeof h);
    h[0] = 0x45; h[1] = 0x67; h[2] = 0x89; h[3] = 0xab; h[5] = 1;
    put_be64(h + 8, (uint64_t)vol * site_bytes);
    strncpy((char *)h + 16, "scidac-binary-data", 127);
    ok = ok && fwrite(h, 1, 144, f) == 144;
  }
  std::vector<unsigned char> buf((size_t)site_bytes * lat[0]);
  Checksum cs;
  uint64_t rank = 0;
  int x[4];
  const uint16_t one = 1;
  const bool little = *(const unsigned char *)&one == 1;
  for (x[3] = 0; x[3] < lat[3] && ok; x[3]++)
    for (x[2] = 0; x[2] < lat[2] && ok; x[2]++)
      for (x[1] = 0; x[1] < lat[1] && ok; x[1]++) {
        for (x[0] = 0; x[0] < lat[0]; x[0]++, rank++) {
          unsigned char *s = buf.data() + (size_t)site_bytes * x[0];
          const unsigned char *d = (const unsigned char *)data + eo_index(lat, x) * (size_t)site_bytes;
          for (int w = 0; w < site_bytes; w += word_bytes)
            for (int b = 0; b < word_bytes; b++) s[w + b] = little ? d[w + word_bytes - 1 - b] : d[w + b];
          cs.add(s, site_bytes, rank);
        }
        ok = fwrite(buf.data(), 1, buf.size(), f) == buf.size();
      }
  {
    static const unsigned char zero[8] = {0};
    const size_t pad = (8 - ((size_t)vol * site_bytes) % 8) % 8;
    ok = ok && (!pad || fwrite(zero, 1, pad, f) == pad);
  }
  snprintf(xml, sizeof xml,
           "<?xml version=\"1.0\" encoding=\"UTF-8\"?><scidacChecksum><version>1.0</version><suma>%x</suma><sumb>%x</sumb>"
           "</scidacChecksum>", cs.a, cs.b);
  ok = ok && write_record(f, "scidac-checksum", false, true, xml, strlen(xml) + 1);
  ok = (fclose(f) == 0) && ok;
  if (!ok) { qexhip_set_error("scidac: write failed"); return QEXHIP_ERR_ARG; }
  return 0;
}

static int io_read_field_impl(const char *path, const int lat[4], void *data, int site_bytes, int word_bytes, char datatype[64]) {
  if (!path || !lat || !data || site_bytes < 1 || (word_bytes != 4 && word_bytes != 8) || site_bytes % word_bytes) return QEXHIP_ERR_ARG;
  FILE *f = fopen(path, "rb");
  if (!f) { qexhip_set_error("scidac: cannot open file"); return QEXHIP_ERR_ARG; }
  FileInfo I;
  if (scan(f, I)) { fclose(f); return QEXHIP_ERR_ARG; }
  const size_t vol = (size_t)lat[0] * lat[1] * lat[2] * lat[3];
  for (int i = 0; i < 4; i++)
    if (I.lat[i] != lat[i]) { fclose(f); qexhip_set_error("scidac: file lattice differs from the requested one"); return QEXHIP_ERR_ARG; }
  if (I.data_len != vol * (size_t)site_bytes || (I.typesize && I.typesize * I.datacount != site_bytes)) {
    fclose(f);
    qexhip_set_error("scidac: record does not hold %d bytes per site", site_bytes);
    return QEXHIP_ERR_ARG;
  }
  if (datatype) { strncpy(datatype, I.datatype.c_str(), 63); datatype[63] = 0; }
  fseeko(f, I.data_off, SEEK_SET);
  std::vector<unsigned char> buf((size_t)site_bytes * lat[0]);
  Checksum cs;
  uint64_t rank = 0;
  int x[4];
  const uint16_t one = 1;
  const bool little = *(const unsigned char *)&one == 1;
  for (x[3] = 0; x[3] < lat[3]; x[3]++)
    for (x[2] = 0; x[2] < lat[2]; x[2]++)
      for (x[1] = 0; x[1] < lat[1]; x[1]++) {
        if (fread(buf.data(), 1, buf.size(), f) != buf.size()) { fclose(f); qexhip_set_error("scidac: short read"); return QEXHIP_ERR_ARG; }
        for (x[0] = 0; x[0] < lat[0]; x[0]++, rank++) {
          const unsigned char *s = buf.data() + (size_t)site_bytes * x[0];
          cs.add(s, site_bytes, rank);
          unsigned char *d = (unsigned char *)data + eo_index(lat, x) * (size_t)site_bytes;
          for (int w = 0; w < site_bytes; w += word_bytes)
            for (int b = 0; b < word_bytes; b++) d[w + b] = little ? s[w + word_bytes - 1 - b] : s[w + b];
        }
      }
  fclose(f);
  if (I.have_sum && (cs.a != I.suma || cs.b != I.sumb)) { qexhip_set_error("scidac: checksum mismatch"); return QEXHIP_ERR_IO; }
  return 0;
}

// Reader.fileMetadata / Reader.recordMetadata (src/io/readerQiolite.nim:37-68,120-135): the user strings of the first
// record; returns the lengths needed (incl. the terminating 0) when a buffer is too small or NULL
static int io_metadata_impl(const char *path, char *file_md, int file_cap, char *record_md, int record_cap, int *file_len,
                                  int *record_len) {
  if (!path) return QEXHIP_ERR_ARG;
  FILE *f = fopen(path, "rb");
  if (!f) { qexhip_set_error("scidac: cannot open file"); return QEXHIP_ERR_ARG; }
  FileInfo I;
  const int r = scan(f, I);
  fclose(f);
  if (r) return QEXHIP_ERR_ARG;
  // the records hold C strings: drop the trailing 0 the writer adds
  while (!I.file_md.empty() && I.file_md.back() == 0) I.file_md.pop_back();
  while (!I.record_md.empty() && I.record_md.back() == 0) I.record_md.pop_back();
  if (file_len) *file_len = (int)I.file_md.size() + 1;
  if (record_len) *record_len = (int)I.record_md.size() + 1;
  if (file_md && file_cap > 0) { strncpy(file_md, I.file_md.c_str(), file_cap - 1); file_md[file_cap - 1] = 0; }
  if (record_md && record_cap > 0) { strncpy(record_md, I.record_md.c_str(), record_cap - 1); record_md[record_cap - 1] = 0; }
  return 0;
}

// ---- the C boundary: no C++ exception may cross it (a truncated or hostile file must come back as QEXHIP_ERR_IO, not as
// std::terminate inside the caller's process) ----
#include <exception>
#include <new>
template <class F> static int guarded(F f) {
  try {
    return f();
  } catch (const std::bad_alloc &) {
    qexhip_set_error("scidac: out of memory (a record length in the file?)");
  } catch (const std::exception &e) {
    qexhip_set_error("scidac: %s", e.what());
  } catch (...) {
    qexhip_set_error("scidac: unknown exception");
  }
  return QEXHIP_ERR_IO;
}
extern "C" int qexhip_io_gauge_info(const char *path, int lat[4], char *precision, int *checksums_present) { return guarded([&] { return io_gauge_info_impl(path, lat, precision, checksums_present); }); }
extern "C" int qexhip_io_read_gauge(const char *path, const int lat[4], double *g, unsigned *suma, unsigned *sumb) { return guarded([&] { return io_read_gauge_impl(path, lat, g, suma, sumb); }); }
extern "C" int qexhip_io_read_gauge_slab(const char *path, const int lat[4], int t0, int nt, double *g) { return guarded([&] { return io_read_gauge_slab_impl(path, lat, t0, nt, g); }); }
extern "C" int qexhip_io_write_gauge(const char *path, const int lat[4], const double *g, char precision,
                                     const char *file_md, const char *record_md) { return guarded([&] { return io_write_gauge_impl(path, lat, g, precision, file_md, record_md); }); }
extern "C" int qexhip_io_write_field(const char *path, const int lat[4], const void *data, int site_bytes, int word_bytes,
                                     const char *datatype, char precision, int colors, int datacount, const char *file_md,
                                     const char *record_md) { return guarded([&] { return io_write_field_impl(path, lat, data, site_bytes, word_bytes, datatype, precision, colors, datacount, file_md, record_md); }); }
extern "C" int qexhip_io_read_field(const char *path, const int lat[4], void *data, int site_bytes, int word_bytes, char datatype[64]) { return guarded([&] { return io_read_field_impl(path, lat, data, site_bytes, word_bytes, datatype); }); }
extern "C" int qexhip_io_metadata(const char *path, char *file_md, int file_cap, char *record_md, int record_cap, int *file_len,
                                  int *record_len) { return guarded([&] { return io_metadata_impl(path, file_md, file_cap, record_md, record_cap, file_len, record_len); }); }

// the CRC-32 the checksums are built from (crc32 of src/io/crc32.nim:35-37), for hosts that want to verify records themselves
extern "C" int qexhip_io_crc32(const void *data, size_t nbytes, unsigned *crc) {
  if ((!data && nbytes) || !crc) return QEXHIP_ERR_ARG;
  *crc = crc32((const unsigned char *)data, nbytes);
  return 0;
}
