// DECLARATION-ONLY stand-in for the part of OpenCV 3.4's public interface that hySLAM's own headers and this repository's adaptors name.
// Test infrastructure (tests/test_adaptor_typecheck.py): it lets a compiler TYPE-CHECK hyslam_amd/host/*.h with -DHYSLAM_AMD_WITH_HYSLAM against the REAL
// hySLAM headers under /root/reference/src (g++ -fsyntax-only).  Nothing here has a body, nothing is linked, shipped or used for arithmetic; the
// signatures are OpenCV 3.4's (opencv2/core/mat.hpp, types.hpp, persistence.hpp) written from its documented API.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>
#include <iostream>
#include <memory>
#include <cmath>
#include <algorithm>
#include <thread>
#include <mutex>
#include <list>
#include <map>
#include <set>

typedef unsigned char uchar;
typedef unsigned short ushort;
#define CV_8U 0
#define CV_8S 1
#define CV_16U 2
#define CV_16S 3
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_8UC4 CV_MAKETYPE(CV_8U, 4)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_64FC1 CV_MAKETYPE(CV_64F, 1)
#define CV_PI 3.1415926535897932384626433832795

namespace cv {
typedef std::string String;
template <typename T> class Point_ { public: Point_(); Point_(T x_, T y_); T x, y; };
typedef Point_<int> Point2i; typedef Point_<float> Point2f; typedef Point_<double> Point2d; typedef Point2i Point;
template <typename T> class Point3_ { public: Point3_(); Point3_(T x_, T y_, T z_); T x, y, z; };
typedef Point3_<float> Point3f; typedef Point3_<double> Point3d;
template <typename T> class Size_ { public: Size_(); Size_(T w, T h); T width, height; };
typedef Size_<int> Size;
template <typename T> class Rect_ { public: Rect_(); Rect_(T x_, T y_, T w, T h); T x, y, width, height; };
typedef Rect_<int> Rect;
template <typename T, int n> class Vec { public: T val[n]; T& operator[](int i); const T& operator[](int i) const; };
template <typename T> class Scalar_ { public: Scalar_(); Scalar_(T v0, T v1 = 0, T v2 = 0, T v3 = 0); T val[4]; };
typedef Scalar_<double> Scalar;
class Range { public: Range(); Range(int s, int e); static Range all(); int start, end; };
template <typename T> using Ptr = std::shared_ptr<T>;

class Mat;
class MatExpr { public: operator Mat() const; MatExpr t() const; MatExpr inv(int method = 0) const; MatExpr mul(const MatExpr& e, double scale = 1) const; };
class _InputArray;
class Mat {
public:
    Mat(); Mat(int rows, int cols, int type); Mat(Size size, int type); Mat(int rows, int cols, int type, const Scalar& s);
    Mat(int rows, int cols, int type, void* data, size_t step = 0); Mat(const Mat& m); Mat(const Mat& m, const Rect& roi); Mat(const MatExpr& e);
    ~Mat();
    Mat& operator=(const Mat& m); Mat& operator=(const MatExpr& e); Mat& operator=(const Scalar& s);
    Mat row(int y) const; Mat col(int x) const; Mat rowRange(int startrow, int endrow) const; Mat colRange(int startcol, int endcol) const;
    Mat clone() const; void copyTo(const _InputArray& m) const; void convertTo(const _InputArray& m, int rtype, double alpha = 1, double beta = 0) const;
    MatExpr t() const; MatExpr inv(int method = 0) const; MatExpr mul(const _InputArray& m, double scale = 1) const;
    double dot(const _InputArray& m) const; Mat cross(const _InputArray& m) const;
    static MatExpr zeros(int rows, int cols, int type); static MatExpr ones(int rows, int cols, int type); static MatExpr eye(int rows, int cols, int type);
    void create(int rows, int cols, int type); void release(); void push_back(const Mat& m);
    Mat operator()(const Rect& roi) const; Mat operator()(Range rowRange, Range colRange) const;
    bool isContinuous() const; size_t elemSize() const; int type() const; int depth() const; int channels() const; bool empty() const; size_t total() const; Size size() const;
    uchar* ptr(int i0 = 0); const uchar* ptr(int i0 = 0) const;
    template <typename T> T* ptr(int i0 = 0); template <typename T> const T* ptr(int i0 = 0) const;
    template <typename T> T& at(int i0, int i1); template <typename T> const T& at(int i0, int i1) const;
    template <typename T> T& at(int i0); template <typename T> const T& at(int i0) const;
    int flags, dims, rows, cols; uchar* data;
    struct MStep { size_t operator[](int i) const; operator size_t() const; size_t* p; size_t buf[2]; } step;
};
template <typename T> class Mat_ : public Mat { public: Mat_(); Mat_(int rows, int cols); Mat_(const Mat& m); T& operator()(int r, int c); const T& operator()(int r, int c) const; };
MatExpr operator+(const Mat& a, const Mat& b); MatExpr operator-(const Mat& a, const Mat& b); MatExpr operator*(const Mat& a, const Mat& b);
MatExpr operator*(const Mat& a, double s); MatExpr operator*(double s, const Mat& a); MatExpr operator/(const Mat& a, double s); MatExpr operator-(const Mat& a);
MatExpr operator+(const MatExpr& a, const Mat& b); MatExpr operator+(const Mat& a, const MatExpr& b); MatExpr operator+(const MatExpr& a, const MatExpr& b);
MatExpr operator-(const MatExpr& a, const Mat& b); MatExpr operator-(const Mat& a, const MatExpr& b); MatExpr operator-(const MatExpr& a, const MatExpr& b);
MatExpr operator*(const MatExpr& a, const Mat& b); MatExpr operator*(const Mat& a, const MatExpr& b); MatExpr operator*(const MatExpr& a, const MatExpr& b);
MatExpr operator*(const MatExpr& a, double s); MatExpr operator*(double s, const MatExpr& a); MatExpr operator/(const MatExpr& a, double s); MatExpr operator-(const MatExpr& a);
std::ostream& operator<<(std::ostream& o, const Mat& m);

class _InputArray {
public:
    _InputArray(); _InputArray(const Mat& m); _InputArray(const MatExpr& e); template <typename T> _InputArray(const std::vector<T>& v);
    Mat getMat(int idx = -1) const; bool empty() const; int type(int i = -1) const; Size size(int i = -1) const;
};
class _OutputArray : public _InputArray { public: _OutputArray(); _OutputArray(Mat& m); template <typename T> _OutputArray(std::vector<T>& v); };
typedef const _InputArray& InputArray; typedef const _OutputArray& OutputArray; typedef const _OutputArray& InputOutputArray;
InputArray noArray();

class KeyPoint {
public:
    KeyPoint(); KeyPoint(Point2f _pt, float _size, float _angle = -1, float _response = 0, int _octave = 0, int _class_id = -1);
    KeyPoint(float x, float y, float _size, float _angle = -1, float _response = 0, int _octave = 0, int _class_id = -1);
    Point2f pt; float size; float angle; float response; int octave; int class_id;
};
class DMatch { public: DMatch(); int queryIdx, trainIdx, imgIdx; float distance; };

class FileNode {
public:
    FileNode(); FileNode operator[](const String& nodename) const; FileNode operator[](const char* nodename) const; FileNode operator[](int i) const;
    bool empty() const; bool isNone() const; bool isSeq() const; bool isMap() const; bool isInt() const; bool isReal() const; bool isString() const; size_t size() const;
    operator int() const; operator float() const; operator double() const; operator std::string() const; String string() const; Mat mat() const;
};
class FileStorage {
public:
    enum Mode { READ = 0, WRITE = 1, APPEND = 2, MEMORY = 4 };
    FileStorage(); FileStorage(const String& filename, int flags, const String& encoding = String()); ~FileStorage();
    bool open(const String& filename, int flags, const String& encoding = String()); bool isOpened() const; void release();
    FileNode operator[](const String& nodename) const; FileNode operator[](const char* nodename) const; FileNode root(int streamidx = 0) const;
};
void operator>>(const FileNode& n, Mat& m); void operator>>(const FileNode& n, int& v); void operator>>(const FileNode& n, float& v); void operator>>(const FileNode& n, double& v); void operator>>(const FileNode& n, std::string& v);

double norm(InputArray src1, int normType = 4, InputArray mask = noArray());
double norm(InputArray src1, InputArray src2, int normType = 4, InputArray mask = noArray());
void hconcat(InputArray src1, InputArray src2, OutputArray dst); void vconcat(InputArray src1, InputArray src2, OutputArray dst);
int cvRound(double v); int cvFloor(double v); int cvCeil(double v); float fastAtan2(float y, float x);
template <typename T> T saturate_cast(double v);
}  // namespace cv
using cv::cvRound; using cv::cvFloor; using cv::cvCeil;
