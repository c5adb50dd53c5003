// fft.rs — DESTINATION: src/mi355x/fft.rs (module crate::mi355x::fft; the crate-root src/fft.rs stays as it is).
// FFT<T> / RealFFT<T> (src/fft.rs:5-56) on the MI355X library.  The reference is generic over rustfft's FftNum; its only
// instantiations in the crate are f32, which is what the library computes in: the wrappers are generic over a one-impl trait
// so that `FFT::<f32>::new(len)` keeps compiling unchanged.  Lengths: ANY length up to 2^23 — the in-LDS plans
// (gm_fft_supported_sizes), their four-step composites, and Bluestein's algorithm for everything else (a length the library
// refuses, e.g. 0 or beyond 2^23, panics in `execute` like an unsupported rustfft feature would).
use crate::mi355x::*;
use num_complex::Complex;

pub trait GmFftNum: Copy + Default {
    fn c2c(len: usize, data: &mut [Complex<Self>]);
    fn power(len: usize, data: &mut [Complex<Self>]) -> Vec<Self>;
    fn r2c(len: usize, input: &[Self]) -> Vec<Complex<Self>>;
    fn norm_sqr(c: &Complex<Self>) -> Self;
}
impl GmFftNum for f32 {
    fn c2c(len: usize, data: &mut [Complex<f32>]) {
        assert!(data.len() >= len);
        let st = unsafe { gm_fft_c2c_f32(len, 0, data.as_mut_ptr(), 1) };
        assert_eq!(st, 0, "gm_fft_c2c_f32: {}", last_error());
    }
    fn power(len: usize, data: &mut [Complex<f32>]) -> Vec<f32> {
        let mut p = vec![0.0f32; len];
        let st = unsafe { gm_fft_power_spectrum_f32(len, data.as_mut_ptr(), p.as_mut_ptr()) };
        assert_eq!(st, 0, "gm_fft_power_spectrum_f32: {}", last_error());
        p
    }
    fn r2c(len: usize, input: &[f32]) -> Vec<Complex<f32>> {
        let mut out = vec![Complex { re: 0.0f32, im: 0.0f32 }; len / 2 + 1];
        let st = unsafe { gm_rfft_f32(len, input.as_ptr(), out.as_mut_ptr()) };
        assert_eq!(st, 0, "gm_rfft_f32: {}", last_error());
        out
    }
    fn norm_sqr(c: &Complex<f32>) -> f32 { c.re * c.re + c.im * c.im }
}

pub struct FFT<T: GmFftNum> {
    len: usize,
    _t: std::marker::PhantomData<T>,
}

impl<T: GmFftNum> FFT<T> {
    pub fn new(len: usize) -> Self {
        Self { len, _t: std::marker::PhantomData }
    }

    pub fn execute(&self, input: &mut [Complex<T>]) -> Vec<Complex<T>> {
        T::c2c(self.len, input);            // in place, like rustfft's process(); then the reference's to_vec()
        input.to_vec()
    }

    pub fn power_spectrum(&self, input: &mut [Complex<T>]) -> Vec<T> {
        T::power(self.len, input)           // transforms `input` in place and returns |X|^2 (:27-29)
    }
}

pub struct RealFFT<T: GmFftNum> {
    len: usize,
    _t: std::marker::PhantomData<T>,
}

impl<T: GmFftNum> RealFFT<T> {
    pub fn new(len: usize) -> Self {
        Self { len, _t: std::marker::PhantomData }
    }

    pub fn execute(&self, input: &mut [T]) -> Vec<Complex<T>> {
        T::r2c(self.len, input)             // len/2 + 1 bins (:45-49)
    }

    pub fn power_spectrum(&self, input: &mut [T]) -> Vec<T> {
        self.execute(input).iter().map(|c| T::norm_sqr(c)).collect()
    }
}
