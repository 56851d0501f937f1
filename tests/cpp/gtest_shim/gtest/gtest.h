// A minimal stand-in for <gtest/gtest.h> (googletest is not installed here): just enough to compile the reference's
// RadioNode/src/test/TEST_*.cpp UNCHANGED against the drop-in headers under include/compat and run them.  It provides
// TEST, ASSERT_EQ / ASSERT_NEAR / ASSERT_TRUE / ASSERT_FALSE with `<<` message sinks, and a main() (SDRX_GTEST_MAIN)
// that runs every registered test (optional argument: a substring filter) and returns the number of failures.
// Test scaffolding for the drop-in boundary only -- no part of the product or of the oracle.
#ifndef SDRX_GTEST_SHIM_H
#define SDRX_GTEST_SHIM_H

#include <cmath>
#include <cstdio>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

namespace testing {
struct TestInfo {
    const char *suite, *name;
    void (*fn)();
};
inline std::vector<TestInfo> &registry()
{
    static std::vector<TestInfo> r;
    return r;
}
inline int &failures_in_current_test()
{
    static int n = 0;
    return n;
}
struct Registrar {
    Registrar(const char *s, const char *n, void (*f)()) { registry().push_back(TestInfo{s, n, f}); }
};
struct Message {
    std::ostringstream os;
    template <class T> Message &operator<<(const T &v)
    {
        os << v;
        return *this;
    }
    Message &operator<<(std::ostream &(*m)(std::ostream &))
    {
        os << m;
        return *this;
    }
};
struct AssertHelper {
    const char *file;
    int line;
    std::string what;
    void operator=(const Message &m) const
    {
        failures_in_current_test()++;
        std::printf("%s:%d: Failure\n  %s\n  %s\n", file, line, what.c_str(), m.os.str().c_str());
    }
};
template <class A, class B> inline bool shim_eq(const A &a, const B &b) { return a == b; }
template <class A, class B> inline std::string shim_show(const char *ea, const char *eb, const A &a, const B &b)
{
    std::ostringstream os;
    os << "Expected equality of " << ea << " (" << +a << ") and " << eb << " (" << +b << ")";
    return os.str();
}
inline int run_all(int argc, char **argv)
{
    const char *filter = argc > 1 ? argv[1] : nullptr;
    int ran = 0, failed = 0;
    for (const TestInfo &t : registry()) {
        const std::string full = std::string(t.suite) + "." + t.name;
        if (filter && full.find(filter) == std::string::npos) continue;
        failures_in_current_test() = 0;
        std::printf("[ RUN      ] %s\n", full.c_str());
        std::fflush(stdout);
        t.fn();
        ran++;
        if (failures_in_current_test()) failed++;
        std::printf("[  %s ] %s\n", failures_in_current_test() ? "FAILED" : "    OK", full.c_str());
        std::fflush(stdout);
    }
    std::printf("[==========] %d tests ran, %d failed\n", ran, failed);
    return failed;
}
} // namespace testing

#define SDRX_GTEST_FAIL_(what) return ::testing::AssertHelper{__FILE__, __LINE__, what} = ::testing::Message()

#define TEST(suite, name)                                                                                   \
    static void suite##_##name##_body();                                                                     \
    static ::testing::Registrar suite##_##name##_registrar(#suite, #name, &suite##_##name##_body);          \
    static void suite##_##name##_body()

#define ASSERT_TRUE(c) \
    if (c) ; else SDRX_GTEST_FAIL_(std::string("Value of: " #c "\n    Actual: false\n  Expected: true"))
#define ASSERT_FALSE(c) \
    if (!(c)) ; else SDRX_GTEST_FAIL_(std::string("Value of: " #c "\n    Actual: true\n  Expected: false"))
#define ASSERT_EQ(a, b) \
    if (::testing::shim_eq((a), (b))) ; else SDRX_GTEST_FAIL_(::testing::shim_show(#a, #b, (a), (b)))
#define ASSERT_NEAR(a, b, tol)                                                                                      \
    if (std::fabs((double)(a) - (double)(b)) <= (double)(tol)) ; else                                                \
        SDRX_GTEST_FAIL_(std::string("The difference between " #a " and " #b " exceeds " #tol) + " (" +          \
                         std::to_string((double)(a)) + " vs " + std::to_string((double)(b)) + ")")
#define EXPECT_TRUE ASSERT_TRUE
#define EXPECT_FALSE ASSERT_FALSE
#define EXPECT_EQ ASSERT_EQ
#define EXPECT_NEAR ASSERT_NEAR

#ifdef SDRX_GTEST_MAIN
int main(int argc, char **argv) { return ::testing::run_all(argc, argv) ? 1 : 0; }
#endif

#endif
