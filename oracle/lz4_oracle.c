/*
 * oracle/lz4_oracle.c -- CPU restatement of the reference hot path
 * (LZ4 v1.9.3 block codec as driven by Streamly.Internal.LZ4).
 *
 * TEST INFRASTRUCTURE ONLY -- see lz4_oracle.h.  Not shipped, not linked into
 * the product, never on the measured path except as bench.py's cpu_baseline
 * ("port" kind) when oracle/_ref is unavailable.
 *
 * This is a restatement written from the algorithm, not a copy: plain index
 * arithmetic on byte arrays, one function per reference stage, each citing the
 * reference lines it follows.  Pinned against the compiled reference
 * (oracle/_ref) by tests/test_oracle.py and against tests/golden/.
 */
#include "lz4_oracle.h"

#include <stdlib.h>
#include <string.h>

/* Block-format constants: cbits/lz4.c:214-235, 634; cbits/lz4.h:557 */
enum {
    K_MINMATCH = 4,
    K_LASTLITERALS = 5,
    K_MFLIMIT = 12,
    K_MINLENGTH = 13,
    K_MAXDIST = 65535,
    K_SKIPTRIGGER = 6,
    K_FASTLOOP_SAFE = 64,
    K_MATCH_SAFEGUARD = 12
};

static uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

/* cbits/lz4.h:171, cbits/lz4.c:674 */
int orc_compress_bound(int n)
{
    if ((unsigned)n > (unsigned)ORC_MAX_INPUT_SIZE) return 0;
    return n + n / 255 + 16;
}

/* ======================================================================
 * Decode
 * ====================================================================== */

/* Variable-length field, cbits/lz4.c:1707-1729.  Returns 0 ok, -1 initial
 * overflow, -2 overflow inside the loop. */
static int read_varlen(const uint8_t *src, long *ip, long lencheck, int initialCheck,
                       uint32_t *len)
{
    uint32_t s;
    if (initialCheck && *ip >= lencheck) return -1;
    do {
        s = src[*ip];
        (*ip)++;
        *len += s;
        if (*ip >= lencheck) return -2;
    } while (s == 255);
    return 0;
}

/* Overlap-safe match copy inside the current block.  offset==0 reproduces
 * v1.9.3's behaviour of emitting zero bytes (cbits/lz4.c:2122-2130, 436-447). */
static void copy_match(uint8_t *dst, long op, long from, uint32_t len, uint32_t offset)
{
    uint32_t i;
    if (offset == 0) { memset(dst + op, 0, len); return; }
    for (i = 0; i < len; i++) dst[op + i] = dst[from + i];
}

/* Match that starts in the external dictionary, cbits/lz4.c:1883-1911 / 2075-2100.
 * back = lowPrefix - match (> 0). */
static void copy_match_extdict(uint8_t *dst, long op, uint32_t len, long back,
                               const uint8_t *dict, size_t dictLen)
{
    if ((long)len <= back) {
        memmove(dst + op, dict + dictLen - back, len);
    } else {
        uint32_t rest = len - (uint32_t)back, i;
        memcpy(dst + op, dict + dictLen - back, (size_t)back);
        op += back;
        for (i = 0; i < rest; i++) dst[op + i] = dst[i]; /* continues from block start */
    }
}

int orc_decompress_safe_dict(const uint8_t *src, int srcLen, uint8_t *dst, int cap,
                             const uint8_t *dict, size_t dictLen)
{
    const long iend = srcLen, oend = cap;
    long ip = 0, op = 0;
    const int useDict = (dict != NULL && dictLen > 0);
    const int checkOffset = (dictLen < 65536); /* cbits/lz4.c:1764 */
    int fast;                                  /* which of the two reference loops we are in */

    if (src == NULL) return -1;                 /* :1752 */
    if (cap == 0) return (srcLen == 1 && src[0] == 0) ? 0 : -1; /* :1781-1785 */
    if (srcLen == 0) return -1;                 /* :1787 */

    fast = (oend - op) >= K_FASTLOOP_SAFE;      /* :1791 */

    for (;;) {
        uint32_t token, ll, ml, offset;
        long cpy, match;
        int haveMatchInfo = 0;

        token = src[ip++];
        ll = token >> 4;

        if (fast) {
            /* ---- fast loop, :1797-1924 ---- */
            if (ll == 15) {
                int e = read_varlen(src, &ip, iend - 15, 1, &ll);      /* :1809 */
                if (e == -1) goto error;                                /* :1810 */
                cpy = op + ll;
                if (cpy > oend - 32 || ip + (long)ll > iend - 32) { fast = 0; goto safe_literal_copy; } /* :1818 */
            } else {
                cpy = op + ll;
                if (ip > iend - 17) { fast = 0; goto safe_literal_copy; } /* :1831 */
            }
            memmove(dst + op, src + ip, ll);
            ip += ll; op = cpy;
            offset = (uint32_t)src[ip] | ((uint32_t)src[ip + 1] << 8); ip += 2;  /* :1844 */
            match = op - (long)offset;
            ml = token & 15;
            if (ml == 15) {
                if (checkOffset && match + (long)dictLen < 0) goto error;   /* :1853 */
                if (read_varlen(src, &ip, iend - K_LASTLITERALS + 1, 0, &ml) != 0) goto error; /* :1854-1855 */
                ml += K_MINMATCH;
                if (op + (long)ml >= oend - K_FASTLOOP_SAFE) { fast = 0; goto safe_match_copy; } /* :1858 */
            } else {
                ml += K_MINMATCH;
                if (op + (long)ml >= oend - K_FASTLOOP_SAFE) { fast = 0; goto safe_match_copy; } /* :1863 */
                if (match >= 0 && offset >= 8) {                           /* :1868-1879 */
                    copy_match(dst, op, match, ml, offset);
                    op += ml;
                    continue;
                }
            }
            if (checkOffset && match + (long)dictLen < 0) goto error;      /* :1881 */
            if (useDict && match < 0) {                                     /* :1883 */
                if (op + (long)ml > oend - K_LASTLITERALS) goto error;      /* :1884-1889 */
                copy_match_extdict(dst, op, ml, -match, dict, dictLen);
                op += ml;
                continue;
            }
            if (match < 0) goto error; /* dictLen>=64K && !useDict cannot happen; defensive */
            copy_match(dst, op, match, ml, offset);                         /* :1914-1923 */
            op += ml;
            continue;
        }

        /* ---- safe loop, :1929-2151 ---- */
        if (ll != 15 && ip < iend - 16 && op <= oend - 32) {               /* shortcut :1944-1974 */
            memmove(dst + op, src + ip, ll);
            op += ll; ip += ll;
            ml = token & 15;
            offset = (uint32_t)src[ip] | ((uint32_t)src[ip + 1] << 8); ip += 2;
            match = op - (long)offset;
            if (ml != 15 && offset >= 8 && match >= 0) {                   /* :1959-1969 */
                copy_match(dst, op, match, ml + K_MINMATCH, offset);
                op += ml + K_MINMATCH;
                continue;
            }
            haveMatchInfo = 1;
            goto copy_match_label;                                           /* :1973 */
        }
        if (ll == 15) {
            int e = read_varlen(src, &ip, iend - 15, 1, &ll);              /* :1979 */
            if (e == -1) goto error;                                        /* :1980 */
        }
        cpy = op + ll;
    safe_literal_copy:
        if (cpy > oend - K_MFLIMIT || ip + (long)ll > iend - (2 + 1 + K_LASTLITERALS)) { /* :1991 */
            if (ip + (long)ll != iend || cpy > oend) goto error;            /* :2031-2036 */
            memmove(dst + op, src + ip, ll);
            ip += ll; op += ll;
            break;                                                          /* :2046 */
        }
        memmove(dst + op, src + ip, ll);                                    /* :2050 */
        ip += ll; op = cpy;
        offset = (uint32_t)src[ip] | ((uint32_t)src[ip + 1] << 8); ip += 2; /* :2055 */
        match = op - (long)offset;
        ml = token & 15;
    copy_match_label:
        (void)haveMatchInfo;
        if (ml == 15) {
            if (read_varlen(src, &ip, iend - K_LASTLITERALS + 1, 0, &ml) != 0) goto error; /* :2064-2065 */
        }
        ml += K_MINMATCH;
    safe_match_copy:
        if (checkOffset && match + (long)dictLen < 0) goto error;           /* :2073 */
        if (useDict && match < 0) {                                         /* :2075 */
            if (op + (long)ml > oend - K_LASTLITERALS) goto error;          /* :2076-2079 */
            copy_match_extdict(dst, op, ml, -match, dict, dictLen);
            op += ml;
            continue;
        }
        if (match < 0) goto error;
        cpy = op + ml;
        if (cpy > oend - K_MATCH_SAFEGUARD) {                               /* :2137 */
            if (cpy > oend - K_LASTLITERALS) goto error;                    /* :2139 */
        }
        copy_match(dst, op, match, ml, offset);
        op = cpy;
    }
    return (int)op;                                                         /* :2156 */

error:
    return (int)(-ip) - 1;                                                  /* :2163 */
}

void orc_dstream_init(orc_dstream *s) { s->prevOut = NULL; s->prevLen = 0; }

/* cbits/lz4.c:2322-2359 under the Haskell call sequence (every block decoded
 * into a fresh allocation => first block noDict, later blocks forceExtDict). */
int orc_decompress_safe_continue(orc_dstream *s, const uint8_t *src, int srcLen,
                                 uint8_t *dst, int cap)
{
    int r;
    if (s->prevLen == 0) {
        r = orc_decompress_safe_dict(src, srcLen, dst, cap, NULL, 0);      /* :2327-2333 */
    } else {
        r = orc_decompress_safe_dict(src, srcLen, dst, cap, s->prevOut, s->prevLen); /* :2347-2355 */
    }
    if (r <= 0) return r;
    s->prevOut = dst;
    s->prevLen = (size_t)r;
    return r;
}

/* ======================================================================
 * Compress
 * ====================================================================== */

/* cbits/lz4.c:706-716 (little-endian branch), hashLog 12 for byU32 */
static uint32_t hash5(const uint8_t *p)
{
    return (uint32_t)(((rd64(p) << 24) * 889523592379ULL) >> (64 - 12));
}

/* cbits/lz4.c:603-626: common-prefix length of a[0..) and b[0..), a bounded by lim */
static uint32_t common_len(const uint8_t *a, const uint8_t *b, const uint8_t *lim)
{
    const uint8_t *s = a;
    while (a + 8 <= lim) {
        uint64_t d = rd64(a) ^ rd64(b);
        if (d) return (uint32_t)(a - s) + (uint32_t)(__builtin_ctzll(d) >> 3);
        a += 8; b += 8;
    }
    while (a < lim && *a == *b) { a++; b++; }
    return (uint32_t)(a - s);
}

void orc_cstream_init(orc_cstream *s) { memset(s, 0, sizeof(*s)); }

/* cbits/lz4.c:1545-1562 */
static long g_renorms;                       /* how often the rebase below has run (tests/test_oracle_renorm.py) */
long orc_debug_renorms(void) { return g_renorms; }
static void renorm(orc_cstream *s, int nextSize)
{
    if (s->currentOffset + (uint32_t)nextSize > 0x80000000u) {
        g_renorms++;
        uint32_t delta = s->currentOffset - 65536u;
        const uint8_t *dictEnd = s->dict + s->dictSize;
        int i;
        for (i = 0; i < 4096; i++)
            s->table[i] = (s->table[i] < delta) ? 0 : s->table[i] - delta;
        s->currentOffset = 65536u;
        if (s->dictSize > 65536u) s->dictSize = 65536u;
        s->dict = dictEnd - s->dictSize;
    }
}

/* Emit helpers (cbits/lz4.c:1033-1046, 1123-1135, 1220-1228) */
static uint8_t *put_litlen(uint8_t *op, uint8_t *token, uint32_t lit)
{
    if (lit >= 15) {
        uint32_t rest = lit - 15;
        *token = 15u << 4;
        while (rest >= 255) { *op++ = 255; rest -= 255; }
        *op++ = (uint8_t)rest;
    } else {
        *token = (uint8_t)(lit << 4);
    }
    return op;
}

/* cbits/lz4.c:851-1240 instantiated (limitedOutput, byU32, usingExtDict, dictIssue) */
static int compress_extdict(orc_cstream *s, const uint8_t *src, uint8_t *dst, int n, int cap,
                            int dictSmall, int accel)
{
    const uint32_t startIndex = s->currentOffset;
    const uint8_t *dict = s->dict;
    const uint32_t dictSize = s->dictSize;
    const uint32_t prefixIdxLimit = startIndex - dictSize;
    const uint8_t *dictEnd = dict ? dict + dictSize : dict;
    const long mflimitPlusOne = (long)n - K_MFLIMIT + 1;
    const uint8_t *matchlimit = src + n - K_LASTLITERALS;
    uint8_t *op = dst, *olimit = dst + cap, *token;
    long ip = 0, anchor = 0;
    uint32_t forwardH, offset = 0;
    const uint8_t *match = NULL;
    int matchInDict = 0;

    s->dictSize += (uint32_t)n;           /* :916 */
    s->currentOffset += (uint32_t)n;      /* :918 */

    if (n < K_MINLENGTH) goto last_literals;  /* :921 */

    s->table[hash5(src)] = startIndex;        /* :924 */
    ip = 1; forwardH = hash5(src + ip);

    for (;;) {
        /* ---- search, :956-1014 ---- */
        {
            long forwardIp = ip;
            int step = 1;
            int searchMatchNb = accel << K_SKIPTRIGGER;
            for (;;) {
                uint32_t h = forwardH;
                uint32_t current = startIndex + (uint32_t)forwardIp;
                uint32_t matchIndex = s->table[h];
                ip = forwardIp;
                forwardIp += step;
                step = (searchMatchNb++ >> K_SKIPTRIGGER);
                if (forwardIp > mflimitPlusOne) goto last_literals;   /* :969 */
                if (matchIndex < startIndex) {                         /* :985-989 */
                    match = dict + (matchIndex - prefixIdxLimit);
                    matchInDict = 1;
                } else {                                               /* :990-993 */
                    match = src + (matchIndex - startIndex);
                    matchInDict = 0;
                }
                forwardH = hash5(src + forwardIp);                     /* :997 */
                s->table[h] = current;                                 /* :998 */
                if (dictSmall && matchIndex < prefixIdxLimit) continue;   /* :1001 */
                if (matchIndex + K_MAXDIST < current) continue;        /* :1003-1006 */
                if (rd32(match) == rd32(src + ip)) {                   /* :1009-1012 */
                    offset = current - matchIndex;
                    break;
                }
            }
        }
        /* ---- catch up, :1019 ---- */
        {
            const uint8_t *low = matchInDict ? dict : src;
            while (ip > anchor && match > low && src[ip - 1] == match[-1]) { ip--; match--; }
        }
        /* ---- literals, :1022-1046 ---- */
        {
            uint32_t lit = (uint32_t)(ip - anchor);
            token = op++;
            if (op + lit + (2 + 1 + K_LASTLITERALS) + (lit / 255) > olimit) return 0;  /* :1024-1027 */
            op = put_litlen(op, token, lit);
            memcpy(op, src + anchor, lit);
            op += lit;
        }
    next_match:
        /* ---- offset, :1065-1068 ---- */
        op[0] = (uint8_t)offset; op[1] = (uint8_t)(offset >> 8); op += 2;
        /* ---- match length, :1076-1136 ---- */
        {
            uint32_t mc;
            if (matchInDict) {                                          /* :1078-1090 */
                const uint8_t *limit = src + ip + (dictEnd - match);
                if (limit > matchlimit) limit = matchlimit;
                mc = common_len(src + ip + K_MINMATCH, match + K_MINMATCH, limit);
                ip += (long)mc + K_MINMATCH;
                if (src + ip == limit) {
                    uint32_t more = common_len(limit, src, matchlimit);
                    mc += more; ip += more;
                }
            } else {                                                    /* :1091-1095 */
                mc = common_len(src + ip + K_MINMATCH, match + K_MINMATCH, matchlimit);
                ip += (long)mc + K_MINMATCH;
            }
            if (op + (1 + K_LASTLITERALS) + (mc + 240) / 255 > olimit) return 0;  /* :1097-1121 */
            if (mc >= 15) {                                             /* :1123-1135 */
                *token += 15;
                mc -= 15;
                while (mc >= 255) { *op++ = 255; mc -= 255; }
                *op++ = (uint8_t)mc;
            } else {
                *token += (uint8_t)mc;
            }
        }
        anchor = ip;
        if (ip >= mflimitPlusOne) break;                                /* :1143 */
        s->table[hash5(src + ip - 2)] = startIndex + (uint32_t)(ip - 2);  /* :1146 */
        /* ---- immediate re-test at ip, :1159-1196 ---- */
        {
            uint32_t h = hash5(src + ip);
            uint32_t current = startIndex + (uint32_t)ip;
            uint32_t matchIndex = s->table[h];
            if (matchIndex < startIndex) { match = dict + (matchIndex - prefixIdxLimit); matchInDict = 1; }
            else { match = src + (matchIndex - startIndex); matchInDict = 0; }
            s->table[h] = current;
            if ((dictSmall ? (matchIndex >= prefixIdxLimit) : 1)
                && (matchIndex + K_MAXDIST >= current)
                && rd32(match) == rd32(src + ip)) {
                token = op++;
                *token = 0;
                offset = current - matchIndex;
                goto next_match;
            }
        }
        forwardH = hash5(src + (++ip));                                 /* :1200 */
    }

last_literals:                                                          /* :1204-1231 */
    {
        uint32_t lastRun = (uint32_t)(n - anchor);
        if (op + lastRun + 1 + ((lastRun + 255 - 15) / 255) > olimit) return 0;
        token = op++;
        op = put_litlen(op, token, lastRun);
        memcpy(op, src + anchor, lastRun);
        op += lastRun;
    }
    return (int)(op - dst);
}

/* cbits/lz4.c:1565-1637, external-dictionary branch only (blocks live in
 * separate allocations under the Haskell call sequence, Internal/LZ4.hs:376,389;
 * the prefix branch :1600-1605 needs dictEnd == src and is not restated). */
int orc_compress_fast_continue(orc_cstream *s, const uint8_t *src, uint8_t *dst, int n, int cap,
                               int accel)
{
    const uint8_t *dictEnd;
    int r, dictSmall;

    if ((uint32_t)n > (uint32_t)ORC_MAX_INPUT_SIZE) return 0;        /* :1262 */
    renorm(s, n);                                                      /* :1576 */
    if (accel < 1) accel = 1;                                          /* :1577 */
    if (accel > 65537) accel = 65537;                                  /* :1578 */

    dictEnd = s->dict + s->dictSize;
    if ((s->dictSize - 1u < 4u - 1u) && dictEnd != src) {             /* :1581-1587 */
        s->dictSize = 0;
        s->dict = src;
        dictEnd = src;
    }
    {   const uint8_t *sourceEnd = src + n;                            /* :1590-1597 */
        if (sourceEnd > s->dict && sourceEnd < dictEnd) {
            s->dictSize = (uint32_t)(dictEnd - sourceEnd);
            if (s->dictSize > 65536u) s->dictSize = 65536u;
            if (s->dictSize < 4) s->dictSize = 0;
            s->dict = dictEnd - s->dictSize;
        }
    }
    if (n == 0) {                                                      /* :1263-1273 */
        if (cap <= 0) return 0;
        dst[0] = 0;
        r = 1;
    } else {
        dictSmall = (s->dictSize < 65536u) && (s->dictSize < s->currentOffset);  /* :1627 */
        r = compress_extdict(s, src, dst, n, cap, dictSmall, accel);
    }
    s->dict = src;                                                     /* :1633 */
    s->dictSize = (uint32_t)n;                                         /* :1634 */
    return r;
}

int orc_compress_block(const uint8_t *src, uint8_t *dst, int n, int cap, int accel)
{
    orc_cstream *s = (orc_cstream *)malloc(sizeof(*s));
    int r;
    if (!s) return 0;
    orc_cstream_init(s);
    r = orc_compress_fast_continue(s, src, dst, n, cap, accel);
    free(s);
    return r;
}

/* ======================================================================
 * Framing: src/Streamly/Internal/LZ4.hs:177-207 (header layout),
 * :226-281 (compressChunk), :291-336 (decompressChunk)
 * ====================================================================== */

static void put_le32(uint8_t *p, int32_t v)
{
    p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}
static int32_t get_le32(const uint8_t *p)
{
    return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
}

size_t orc_frame_stream_compress(const uint8_t *in, size_t inLen, int blockLen, int accel,
                                 int headerKind, int linked, uint8_t *out, size_t outCap)
{
    orc_cstream *s = (orc_cstream *)malloc(sizeof(*s));
    size_t pos = 0, o = 0;
    uint8_t *prev = NULL;
    if (!s) return (size_t)-1;
    orc_cstream_init(s);
    if (accel < 0) accel = 0;                            /* speed = max speed0 0, Internal/LZ4.hs:364 */
    while (pos < inLen) {
        int n = (int)((inLen - pos < (size_t)blockLen) ? inLen - pos : (size_t)blockLen);
        int bound = orc_compress_bound(n), c;
        /* each block lives in its own allocation, like a Haskell Array */
        uint8_t *blk = (uint8_t *)malloc((size_t)n + 8);
        memcpy(blk, in + pos, (size_t)n);
        if (!linked) orc_cstream_init(s);
        if (o + (size_t)headerKind + (size_t)bound > outCap) { free(blk); free(prev); free(s); return (size_t)-1; }
        c = orc_compress_fast_continue(s, blk, out + o + headerKind, n, bound, accel);
        if (c <= 0) { free(blk); free(prev); free(s); return (size_t)-1; }
        put_le32(out + o, c);                            /* Internal/LZ4.hs:262 */
        if (headerKind == 8) put_le32(out + o + 4, n);   /* Internal/LZ4.hs:261 */
        o += (size_t)headerKind + (size_t)c;
        pos += (size_t)n;
        free(prev);                                      /* previous input kept alive one step, :389 */
        prev = blk;
    }
    free(prev);
    free(s);
    return o;
}

size_t orc_frame_stream_decompress(const uint8_t *in, size_t inLen, int headerKind, int fixedUncomp,
                                   int linked, uint8_t *out, size_t outCap)
{
    orc_dstream ds;
    size_t pos = 0, o = 0, k = 0;
    uint8_t *prev = NULL;
    orc_dstream_init(&ds);
    while (pos + (size_t)headerKind <= inLen) {
        int32_t c = get_le32(in + pos);
        int32_t u = (headerKind == 8) ? get_le32(in + pos + 4) : fixedUncomp;
        uint8_t *blk;
        int r;
        if (c <= 0 || pos + (size_t)headerKind + (size_t)c > inLen || u < 0) { free(prev); return (size_t)-1 - k; }
        blk = (uint8_t *)malloc((size_t)u + 8);
        if (!linked) orc_dstream_init(&ds);
        r = orc_decompress_safe_continue(&ds, in + pos + headerKind, c, blk, u);
        if (r < 0 || o + (size_t)r > outCap) { free(blk); free(prev); return (size_t)-1 - k; }
        memcpy(out + o, blk, (size_t)r);
        o += (size_t)r;
        pos += (size_t)headerKind + (size_t)c;
        free(prev);                                      /* previous OUTPUT kept alive one step, :564 */
        prev = blk;
        k++;
    }
    free(prev);
    return o;
}

/* ======================================================================
 * Generators.  Per-block seeds so any device can make any block.
 * ====================================================================== */

static uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
static uint64_t block_seed(uint64_t blockIndex)
{
    uint64_t s = splitmix64(0x9E3779B97F4A7C15ULL ^ blockIndex);
    return s ? s : 1;
}
static uint64_t xs64(uint64_t *st)
{
    uint64_t x = *st;
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    *st = x;
    return x * 0x2545F4914F6CDD1DULL;
}

void orc_gen_random(uint8_t *dst, size_t blockLen, uint64_t blockIndex)
{
    uint64_t st = block_seed(blockIndex);
    size_t i = 0;
    while (i < blockLen) {
        uint64_t r = xs64(&st);
        int k;
        for (k = 0; k < 8 && i < blockLen; k++, i++) dst[i] = (uint8_t)(r >> (8 * k));
    }
}

void orc_gen_lzsynth(uint8_t *dst, size_t blockLen, uint64_t blockIndex, uint32_t litMax, uint32_t offMax)
{
    uint64_t st = block_seed(blockIndex);
    size_t pos = 0;
    while (pos < blockLen) {
        uint32_t L = 1 + (uint32_t)(xs64(&st) % litMax), M, o, lim, i;
        for (i = 0; i < L && pos < blockLen; i++) dst[pos++] = (uint8_t)(32 + xs64(&st) % 64);
        if (pos >= blockLen) break;
        M = 4 + (uint32_t)(xs64(&st) % 61);
        lim = (pos < offMax) ? (uint32_t)pos : offMax;
        o = 1 + (uint32_t)(xs64(&st) % lim);
        for (i = 0; i < M && pos < blockLen; i++, pos++) dst[pos] = dst[pos - o];
    }
}

/* Text-like stand-in (Canterbury is not available offline): words drawn from a
 * 4096-word vocabulary with a skewed (four-factor product) distribution.  The vocabulary is
 * a pure function of the word index, so it is shared by all blocks. */
void orc_gen_text(uint8_t *dst, size_t blockLen, uint64_t blockIndex)
{
    uint64_t st = block_seed(blockIndex ^ 0x7465787400000000ULL);
    size_t pos = 0;
    while (pos < blockLen) {
        uint64_t r = xs64(&st);
        uint32_t a = (uint32_t)(r & 4095), b = (uint32_t)((r >> 12) & 4095), c = (uint32_t)((r >> 29) & 4095), d = (uint32_t)((r >> 41) & 4095);
        uint32_t w = (((a * b) >> 12) * ((c * d) >> 12)) >> 12;   /* quartic skew: a few words dominate */
        uint64_t h = splitmix64(0x776F7264ULL + w);
        uint32_t len = 2 + (uint32_t)(h & 7), j;
        uint32_t sep = (uint32_t)((r >> 24) & 31);
        for (j = 0; j < len && pos < blockLen; j++)
            dst[pos++] = (uint8_t)('a' + ((h >> (3 + 5 * j)) & 31) % 26);
        if (pos < blockLen) dst[pos++] = (sep == 0) ? '\n' : (sep == 1) ? ',' : ' ';
        if (sep == 1 && pos < blockLen) dst[pos++] = ' ';
    }
}
