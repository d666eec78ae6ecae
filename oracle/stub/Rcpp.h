// oracle/stub/Rcpp.h -- TEST INFRASTRUCTURE: just enough of the Rcpp surface for image.CannyEdges/src/rcpp_canny.cpp to
// compile unmodified without R: IntegerVector (size, []), NumericMatrix(Dimension), List::create(_["name"] = value).
// Written for this repo; not derived from Rcpp's sources.
#pragma once
#include <stddef.h>

#include <string>
#include <utility>
#include <vector>

namespace Rcpp {

class IntegerVector {
  public:
    IntegerVector() {}
    IntegerVector(const int *p, size_t n) : v_(p, p + n) {}
    long size() const { return (long)v_.size(); }
    int &operator[](long i) { return v_[(size_t)i]; }
    const int &operator[](long i) const { return v_[(size_t)i]; }
  private:
    std::vector<int> v_;
};

struct Dimension {
    size_t d0, d1;
    Dimension(size_t a, size_t b) : d0(a), d1(b) {}
};

class NumericMatrix {
  public:
    NumericMatrix() : nr_(0), nc_(0) {}
    explicit NumericMatrix(const Dimension &d) : v_(d.d0 * d.d1, 0.0), nr_(d.d0), nc_(d.d1) {}
    double &operator[](long i) { return v_[(size_t)i]; }
    const std::vector<double> &data() const { return v_; }
    size_t nrow() const { return nr_; }
    size_t ncol() const { return nc_; }
  private:
    std::vector<double> v_;
    size_t nr_, nc_;
};

// one named element of a List: either a matrix or a scalar (everything scalar the reference stores fits a double)
struct NamedValue {
    std::string name;
    bool is_matrix = false;
    NumericMatrix matrix;
    double scalar = 0;
};

struct NameProxy {
    std::string name;
    NamedValue operator=(const NumericMatrix &m) const { NamedValue v; v.name = name; v.is_matrix = true; v.matrix = m; return v; }
    template <typename T> NamedValue operator=(const T &x) const { NamedValue v; v.name = name; v.scalar = (double)x; return v; }
};

struct NameMaker {
    NameProxy operator[](const char *n) const { return NameProxy{n}; }
};
static const NameMaker _ = NameMaker();

class List {
  public:
    template <typename... A> static List create(const A &...a)
    {
        List l;
        const NamedValue vals[] = {a...};
        for (const NamedValue &v : vals) l.items_.push_back(v);
        return l;
    }
    const NamedValue &get(const std::string &name) const
    {
        for (const NamedValue &v : items_)
            if (v.name == name) return v;
        static const NamedValue none;
        return none;
    }
  private:
    std::vector<NamedValue> items_;
};

}  // namespace Rcpp
