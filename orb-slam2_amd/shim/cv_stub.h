// Minimal stand-in for the few OpenCV 2.4 core types the ORBextractor / ORBmatcher interfaces
// mention, so that the drop-in classes can be compiled and tested where OpenCV is not installed.
// With real OpenCV, do not define ORBHIP_SHIM_CV_STUB and this header is not used.
#ifndef ORBHIP_CV_STUB_H
#define ORBHIP_CV_STUB_H
#include <cstdint>
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5

namespace cv {

struct Point2f {
    float x, y;
};

struct KeyPoint {  // field-for-field the 2.4 class: 28 bytes
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
};

class Mat;
struct MatT_;    // marker of A.t()
struct MatMul_;  // lazy product, so that A*x + t evaluates as one gemm call the way cv::MatExpr does

// Two element types: CV_8U (images, descriptors) and CV_32F (poses, points).  Views made by row / col /
// rowRange / colRange share the owner's storage.
class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    unsigned char* data = nullptr;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void* ext, size_t st) : rows(r), cols(c), step(st), data((unsigned char*)ext), type_(type) {}
    void create(int r, int c, int type) {
        if (r == rows && c == cols && type == type_ && owner_) return;
        type_ = type;
        const size_t bytes = (size_t)r * c * elemSize();
        owner_.reset(new unsigned char[bytes ? bytes : 1], std::default_delete<unsigned char[]>());
        data = owner_.get();
        rows = r;
        cols = c;
        step = (size_t)c * elemSize();
    }
    void release() {
        owner_.reset();
        data = nullptr;
        rows = cols = 0;
        step = 0;
    }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    bool isContinuous() const { return step == (size_t)cols * elemSize(); }
    int type() const { return type_; }
    size_t elemSize() const { return type_ == CV_32F ? 4 : 1; }
    template <class T> T* ptr(int r = 0) { return (T*)(data + (size_t)r * step); }
    template <class T> const T* ptr(int r = 0) const { return (const T*)(data + (size_t)r * step); }
    template <class T> T& at(int r, int c) { return ((T*)(data + (size_t)r * step))[c]; }
    template <class T> const T& at(int r, int c) const { return ((const T*)(data + (size_t)r * step))[c]; }
    // single index: element i of a row or column vector
    template <class T> T& at(int i) { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
    template <class T> const T& at(int i) const { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
    Mat view(int r0, int r1, int c0, int c1) const {
        Mat v(r1 - r0, c1 - c0, type_, data + (size_t)r0 * step + (size_t)c0 * elemSize(), step);
        v.owner_ = owner_;
        return v;
    }
    Mat row(int r) const { return view(r, r + 1, 0, cols); }
    Mat col(int c) const { return view(0, rows, c, c + 1); }
    Mat rowRange(int a, int b) const { return view(a, b, 0, cols); }
    Mat colRange(int a, int b) const { return view(0, rows, a, b); }
    Mat clone() const {
        Mat o(rows, cols, type_);
        for (int r = 0; r < rows; r++) memcpy(o.data + (size_t)r * o.step, data + (size_t)r * step, (size_t)cols * elemSize());
        return o;
    }
    // Mat::dot: accumulates in double
    double dot(const Mat& b) const {
        double r = 0;
        for (int i = 0; i < rows; i++)
            for (int j = 0; j < cols; j++) r += (double)at<float>(i, j) * b.at<float>(i, j);
        return r;
    }
    inline MatT_ t() const;
    inline Mat(const MatMul_& e);

private:
    std::shared_ptr<unsigned char> owner_;
    int type_ = CV_8U;
};

struct MatT_ {
    Mat a;
    double alpha;
};
struct MatMul_ {
    Mat a, b;
    bool ta;
    double alpha;
};
struct MatScaled_ {  // -A: a matrix with a scale that a following product folds into its alpha
    Mat a;
    double alpha;
};
inline MatT_ Mat::t() const { return MatT_{*this, 1.0}; }
inline MatT_ operator-(const MatT_& t) { return MatT_{t.a, -t.alpha}; }
inline MatScaled_ operator-(const Mat& a) { return MatScaled_{a, -1.0}; }
inline MatMul_ operator*(const Mat& a, const Mat& b) { return MatMul_{a, b, false, 1.0}; }
inline MatMul_ operator*(const MatT_& a, const Mat& b) { return MatMul_{a.a, b, true, a.alpha}; }
inline MatMul_ operator*(const MatScaled_& a, const Mat& b) { return MatMul_{a.a, b, false, a.alpha}; }
// s * M and s * M.t(): convertTo with scale s (after the transpose), i.e. a float multiply by (float)s
inline Mat operator*(double s, const Mat& a) {
    const float f = (float)s;
    Mat o(a.rows, a.cols, CV_32F);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < a.cols; j++) o.at<float>(i, j) = a.at<float>(i, j) * f;
    return o;
}
inline Mat operator*(double s, const MatT_& t) {
    const float f = (float)(s * t.alpha);
    Mat o(t.a.cols, t.a.rows, CV_32F);
    for (int i = 0; i < t.a.cols; i++)
        for (int j = 0; j < t.a.rows; j++) o.at<float>(i, j) = t.a.at<float>(j, i) * f;
    return o;
}

// cv::gemm as OpenCV 2.4 evaluates the two shapes the matchers use (CV_32F): a plain product of small
// matrices accumulates each element in float, left to right, and finishes as (float)(acc*alpha + c*beta) with
// double scale factors; a product with a transposed first factor takes the general path, which accumulates
// in double, multiplies by alpha and rounds once.
inline Mat gemm_(const MatMul_& e, const Mat* c) {
    const int M = e.ta ? e.a.cols : e.a.rows, K = e.ta ? e.a.rows : e.a.cols, N = e.b.cols;
    Mat o(M, N, CV_32F);
    for (int i = 0; i < M; i++)
        for (int j = 0; j < N; j++) {
            double r;
            if (!e.ta) {
                float acc = 0.f;
                for (int k = 0; k < K; k++) {
                    const float p = e.a.at<float>(i, k) * e.b.at<float>(k, j);
                    acc = k ? acc + p : p;
                }
                r = acc * e.alpha;
            } else {
                double acc = 0;
                for (int k = 0; k < K; k++) acc += (double)e.a.at<float>(k, i) * (double)e.b.at<float>(k, j);
                r = acc * e.alpha;
            }
            if (c) r += c->at<float>(i, j) * 1.0;
            o.at<float>(i, j) = (float)r;
        }
    return o;
}
inline Mat::Mat(const MatMul_& e) { *this = gemm_(e, nullptr); }
inline Mat operator+(const MatMul_& e, const Mat& c) { return gemm_(e, &c); }
inline Mat operator-(const Mat& a, const Mat& b) {
    Mat o(a.rows, a.cols, CV_32F);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < a.cols; j++) o.at<float>(i, j) = a.at<float>(i, j) - b.at<float>(i, j);
    return o;
}
// M / s: convertTo with scale 1/s, i.e. a float multiply by (float)(1.0 / s)
inline Mat operator/(const Mat& a, double s) {
    const float f = (float)(1.0 / s);
    Mat o(a.rows, a.cols, CV_32F);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < a.cols; j++) o.at<float>(i, j) = a.at<float>(i, j) * f;
    return o;
}
// cv::norm (L2): double accumulation of squares
inline double norm(const Mat& a) {
    double s = 0;
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < a.cols; j++) s += (double)a.at<float>(i, j) * a.at<float>(i, j);
    return std::sqrt(s);
}

class _InputArray {
public:
    _InputArray(const Mat& m) : m_(&m) {}
    Mat getMat() const { return *m_; }
    bool empty() const { return m_->empty(); }

private:
    const Mat* m_;
};
class _OutputArray {
public:
    _OutputArray(Mat& m) : m_(&m) {}
    void create(int r, int c, int t) const { m_->create(r, c, t); }
    void release() const { m_->release(); }
    Mat getMat() const { return *m_; }

private:
    Mat* m_;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;

}  // namespace cv
#endif
