//! The single-prover proof: `Groth16::<E, CircomReduction>::create_proof_with_reduction_and_matrices`
//! (groth16/examples/sha256.rs:159, mpc-api/src/main.rs:393) as `dg16_qap` + `dg16_groth16_prove` over a key made
//! resident once per circuit (`dg16_pk_create`).
use crate::pack::{pack_affine, scalars_as_bytes, scalars_as_bytes_mut, unpack_projective, FieldBytes};
use crate::{check, sys, Dg16Config, Dg16Error, CTX};
use ark_ec::pairing::Pairing;
use ark_ec::short_weierstrass::{Affine, SWCurveConfig};
use ark_ec::CurveGroup;
use ark_ff::{PrimeField, Zero};
use ark_groth16::{Proof, ProvingKey};
use ark_relations::r1cs::ConstraintMatrices;
use std::os::raw::c_uint;

/// CSR form of `ConstraintMatrices::{a, b}` (`Vec<Vec<(F, usize)>>`): row_ptr u32[nc + 1], col u32[nnz], coeff F[nnz]
/// (Montgomery, as in memory).
pub struct Csr<F> {
    pub row_ptr: Vec<u32>,
    pub col: Vec<u32>,
    pub coeff: Vec<F>,
}
impl<F: PrimeField> Csr<F> {
    pub fn from_rows(rows: &[Vec<(F, usize)>]) -> Self {
        let nnz: usize = rows.iter().map(|r| r.len()).sum();
        let (mut row_ptr, mut col, mut coeff) = (Vec::with_capacity(rows.len() + 1), Vec::with_capacity(nnz), Vec::with_capacity(nnz));
        row_ptr.push(0u32);
        for r in rows {
            for (v, j) in r {
                col.push(*j as u32);
                coeff.push(*v);
            }
            row_ptr.push(col.len() as u32);
        }
        Self { row_ptr, col, coeff }
    }
}

/// A proving key resident in HBM as window tables (built once per circuit; 6 GB for BN254 at 2^20 wires).
pub struct ResidentKey<E: Pairing> {
    pk: *mut sys::Dg16Pk,
    curve: std::os::raw::c_int,
    num_vars: usize,
    num_inputs: usize,
    domain_size: usize,
    _e: core::marker::PhantomData<E>,
}
unsafe impl<E: Pairing> Send for ResidentKey<E> {}
unsafe impl<E: Pairing> Sync for ResidentKey<E> {}
impl<E: Pairing> Drop for ResidentKey<E> {
    fn drop(&mut self) {
        unsafe { sys::dg16_pk_destroy(self.pk) }
    }
}

impl<E, P1, P2> ResidentKey<E>
where
    E: Pairing<G1Affine = Affine<P1>, G2Affine = Affine<P2>>,
    P1: Dg16Config<ScalarField = E::ScalarField>,
    P2: Dg16Config<ScalarField = E::ScalarField>,
    P1::BaseField: FieldBytes,
    P2::BaseField: FieldBytes,
{
    /// Packs `pk.a_query`, `pk.b_g1_query`, `pk.b_g2_query`, `pk.h_query`, `pk.l_query` and
    /// `alpha_g1 | beta_g1 | delta_g1 | beta_g2 | delta_g2` exactly as groth16/src/proving_key.rs:48-65 selects them
    /// (element 0 of the a / b queries is kept: the library adds it like sha256.rs:211-212 does).
    pub fn from_arkworks(pk: &ProvingKey<E>, num_inputs: usize, domain_size: usize) -> Result<Self, Dg16Error> {
        let num_vars = pk.a_query.len();
        let a = pack_affine(&pk.a_query);
        let b1 = pack_affine(&pk.b_g1_query);
        let b2 = pack_affine(&pk.b_g2_query);
        let h = pack_affine(&pk.h_query);
        let l = pack_affine(&pk.l_query);
        let mut fixed = pack_affine(&[pk.vk.alpha_g1, pk.beta_g1, pk.delta_g1]);
        fixed.extend(pack_affine(&[pk.vk.beta_g2, pk.vk.delta_g2]));
        let mut out = core::ptr::null_mut();
        check(unsafe {
            sys::dg16_pk_create(
                CTX.0, P1::CURVE, num_vars, num_inputs, domain_size, a.as_ptr().cast(), b1.as_ptr().cast(), b2.as_ptr().cast(),
                h.as_ptr().cast(), l.as_ptr().cast(), fixed.as_ptr().cast(), 0, &mut out,
            )
        })?;
        Ok(Self { pk: out, curve: P1::CURVE, num_vars, num_inputs, domain_size, _e: core::marker::PhantomData })
    }

    /// `qap::qap` (groth16/src/qap.rs:44-91): a = A w, b = B w on the constraint rows, the instance rows, c = a o b.
    pub fn qap(&self, m: &ConstraintMatrices<E::ScalarField>, w: &[E::ScalarField])
        -> Result<(Vec<E::ScalarField>, Vec<E::ScalarField>, Vec<E::ScalarField>), Dg16Error> {
        let (ca, cb) = (Csr::from_rows(&m.a), Csr::from_rows(&m.b));
        let z = E::ScalarField::zero();
        let (mut a, mut b, mut c) = (vec![z; self.domain_size], vec![z; self.domain_size], vec![z; self.domain_size]);
        check(unsafe {
            sys::dg16_qap(
                CTX.0, self.curve, m.num_constraints, self.num_inputs, self.num_vars, self.domain_size.trailing_zeros(),
                ca.row_ptr.as_ptr(), ca.col.as_ptr(), scalars_as_bytes(&ca.coeff).as_ptr().cast(),
                cb.row_ptr.as_ptr(), cb.col.as_ptr(), scalars_as_bytes(&cb.coeff).as_ptr().cast(),
                scalars_as_bytes(w).as_ptr().cast(),
                scalars_as_bytes_mut(&mut a).as_mut_ptr().cast(), scalars_as_bytes_mut(&mut b).as_mut_ptr().cast(),
                scalars_as_bytes_mut(&mut c).as_mut_ptr().cast(), sys::DG16_F_SCALARS_MONT, 0,
            )
        })?;
        Ok((a, b, c))
    }

    /// h-polynomial, five MSMs (A, B1, L, H in G1; B in G2), A / B / C assembly (groth16/src/prove.rs:21-136 for a
    /// single party).  Uses all three channels of the context.
    pub fn prove(&self, a: &[E::ScalarField], b: &[E::ScalarField], c: &[E::ScalarField], full_assignment: &[E::ScalarField],
                 r: E::ScalarField, s: E::ScalarField) -> Result<Proof<E>, Dg16Error> {
        let (f1, f2) = (<P1::BaseField as FieldBytes>::BYTES, <P2::BaseField as FieldBytes>::BYTES);
        let mut out = vec![0u8; 3 * f1 + 3 * f2 + 3 * f1];
        let rs = [r, s];
        let flags: c_uint = sys::DG16_F_SCALARS_MONT;
        check(unsafe {
            sys::dg16_groth16_prove(
                CTX.0, self.pk, scalars_as_bytes(a).as_ptr().cast(), scalars_as_bytes(b).as_ptr().cast(),
                scalars_as_bytes(c).as_ptr().cast(), scalars_as_bytes(full_assignment).as_ptr().cast(),
                scalars_as_bytes(&rs).as_ptr().cast(), flags, out.as_mut_ptr().cast(),
            )
        })?;
        let pa = unpack_projective::<P1>(&out[..3 * f1]);
        let pb = unpack_projective::<P2>(&out[3 * f1..3 * f1 + 3 * f2]);
        let pc = unpack_projective::<P1>(&out[3 * f1 + 3 * f2..]);
        Ok(Proof { a: pa.into_affine(), b: pb.into_affine(), c: pc.into_affine() })
    }

    /// The whole of `create_proof_with_reduction_and_matrices(&pk, r, s, &matrices, num_inputs, num_constraints,
    /// &full_assignment)`.
    pub fn create_proof_with_reduction_and_matrices(&self, r: E::ScalarField, s: E::ScalarField,
                                                    matrices: &ConstraintMatrices<E::ScalarField>,
                                                    full_assignment: &[E::ScalarField]) -> Result<Proof<E>, Dg16Error> {
        let (a, b, c) = self.qap(matrices, full_assignment)?;
        self.prove(&a, &b, &c, full_assignment, r, s)
    }
}
