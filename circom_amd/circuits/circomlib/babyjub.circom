/*
    BabyJubjub (twisted Edwards a = 168700, d = 168696 over the bn128 scalar field) in the shape of circomlib's
    babyjub.circom, plus bit-serial scalar multiplications: ScalarMulBits (one division hint per coordinate ON the dependency
    chain) and ScalarMulBitsProj (the same constraints; the hints walk a projective ladder, the inversions of the Z's are
    mutually independent and the lowering batches them).  The circom text of circom_amd/circuits/babyjub.py.
*/
pragma circom 2.0.0;

template BabyAdd() {
    signal input x1;
    signal input y1;
    signal input x2;
    signal input y2;
    signal output xout;
    signal output yout;
    signal beta;
    signal gamma;
    signal delta;
    signal tau;
    var a = 168700;
    var d = 168696;
    beta <== x1 * y2;
    gamma <== y1 * x2;
    delta <== (-a * x1 + y1) * (x2 + y2);
    tau <== beta * gamma;
    xout <-- (beta + gamma) / (1 + d * tau);
    (1 + d * tau) * xout === (beta + gamma);
    yout <-- (delta + a * beta - gamma) / (1 - d * tau);
    (1 - d * tau) * yout === (delta + a * beta - gamma);
}

template BabyDbl() {
    signal input x;
    signal input y;
    signal output xout;
    signal output yout;
    component adder = BabyAdd();
    adder.x1 <== x;
    adder.y1 <== y;
    adder.x2 <== x;
    adder.y2 <== y;
    adder.xout ==> xout;
    adder.yout ==> yout;
}

template BabyCheck() {
    signal input x;
    signal input y;
    signal x2;
    signal y2;
    var a = 168700;
    var d = 168696;
    x2 <== x * x;
    y2 <== y * y;
    a * x2 + y2 === 1 + d * x2 * y2;
}

// out = (sum e[i] 2^i) * P by double-and-add from the most significant bit (the e[i] are checked to be bits)
template ScalarMulBits(n) {
    signal input e[n];
    signal input px;
    signal input py;
    signal output outx;
    signal output outy;
    component check = BabyCheck();
    check.x <== px;
    check.y <== py;
    signal selx[n];
    signal sely[n];
    var accx = 0;
    var accy = 1;
    component dbl[n];
    component add[n];
    for (var i = n - 1; i >= 0; i--) {
        e[i] * (e[i] - 1) === 0;
        dbl[i] = BabyDbl();
        dbl[i].x <== accx;
        dbl[i].y <== accy;
        selx[i] <== e[i] * px;                  // e ? P : (0, 1)
        sely[i] <== e[i] * (py - 1) + 1;
        add[i] = BabyAdd();
        add[i].x1 <== dbl[i].xout;
        add[i].y1 <== dbl[i].yout;
        add[i].x2 <== selx[i];
        add[i].y2 <== sely[i];
        accx = add[i].xout;
        accy = add[i].yout;
    }
    outx <== accx;
    outy <== accy;
}

// dbl-2008-bbjlp on (X : Y : Z): 3M + 4S
function proj_dbl(X, Y, Z) {
    var a = 168700;
    var B = (X + Y) * (X + Y);
    var C = X * X;
    var Dd = Y * Y;
    var E = C * a;
    var F = E + Dd;
    var H = Z * Z;
    var J = F - 2 * H;
    return [(B - C - Dd) * J, F * (E - Dd), F * J];
}

// add-2008-bbjlp with Z2 = 1
function proj_add_affine(X1, Y1, Z1, x2, y2) {
    var a = 168700;
    var d = 168696;
    var B = Z1 * Z1;
    var C = X1 * x2;
    var Dd = Y1 * y2;
    var E = C * Dd * d;
    var F = B - E;
    var G = B + E;
    var T = (X1 + Y1) * (x2 + y2) - C - Dd;
    return [Z1 * F * T, Z1 * G * (Dd - C * a), F * G];
}

template ScalarMulBitsProj(n) {
    signal input e[n];
    signal input px;
    signal input py;
    signal output outx;
    signal output outy;
    component check = BabyCheck();
    check.x <== px;
    check.y <== py;
    signal selx[n];
    signal sely[n];
    signal dblx[n];
    signal dbly[n];
    signal addx[n];
    signal addy[n];
    signal aux[n][2][4];          // beta, gamma, delta, tau of the doubling / the addition of step i
    var a = 168700;
    var d = 168696;
    var P[3] = [0, 1, 1];
    var accx = 0;
    var accy = 1;
    var zi;
    for (var i = n - 1; i >= 0; i--) {
        e[i] * (e[i] - 1) === 0;
        selx[i] <== e[i] * px;
        sely[i] <== e[i] * (py - 1) + 1;
        P = proj_dbl(P[0], P[1], P[2]);
        zi = 1 / P[2];
        dblx[i] <-- P[0] * zi;
        dbly[i] <-- P[1] * zi;
        aux[i][0][0] <== accx * accy;
        aux[i][0][1] <== accy * accx;
        aux[i][0][2] <== (-a * accx + accy) * (accx + accy);
        aux[i][0][3] <== aux[i][0][0] * aux[i][0][1];
        (1 + d * aux[i][0][3]) * dblx[i] === aux[i][0][0] + aux[i][0][1];
        (1 - d * aux[i][0][3]) * dbly[i] === aux[i][0][2] + a * aux[i][0][0] - aux[i][0][1];
        P = proj_add_affine(P[0], P[1], P[2], selx[i], sely[i]);
        zi = 1 / P[2];
        addx[i] <-- P[0] * zi;
        addy[i] <-- P[1] * zi;
        aux[i][1][0] <== dblx[i] * sely[i];
        aux[i][1][1] <== dbly[i] * selx[i];
        aux[i][1][2] <== (-a * dblx[i] + dbly[i]) * (selx[i] + sely[i]);
        aux[i][1][3] <== aux[i][1][0] * aux[i][1][1];
        (1 + d * aux[i][1][3]) * addx[i] === aux[i][1][0] + aux[i][1][1];
        (1 - d * aux[i][1][3]) * addy[i] === aux[i][1][2] + a * aux[i][1][0] - aux[i][1][1];
        accx = addx[i];
        accy = addy[i];
    }
    outx <== accx;
    outy <== accy;
}
